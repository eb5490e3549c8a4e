// capi_tv.hip -- C ABI (include/gmmiv.h): total-variability (i-vector) maths and i-vector scoring.
#include <math.h>
#include <string.h>

#include <map>

#include "ctx.h"
#include "tv_kernels.h"

namespace {

// Small dense host helpers for the O(R^3)-once-per-call pieces (min-divergence factor, PLDA K_n).
bool host_cholesky_upper(int n, const std::vector<double> &a, std::vector<double> &ch)
{
    ch.assign((size_t)n * n, 0.0); // R = Ch^T Ch, Ch upper
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = a[(size_t)i * n + j];
            for (int k = 0; k < i; ++k) s -= ch[(size_t)k * n + i] * ch[(size_t)k * n + j];
            if (i == j) {
                if (!(s > 0.0)) return false;
                ch[(size_t)i * n + i] = sqrt(s);
            } else
                ch[(size_t)i * n + j] = s / ch[(size_t)i * n + i];
        }
    return true;
}

// SPD inverse + log det through the Cholesky factor (host)
bool host_spd_inverse(int n, const std::vector<double> &a, std::vector<double> &inv, double *logdet)
{
    std::vector<double> u;
    if (!host_cholesky_upper(n, a, u)) return false;
    double ld = 0.0;
    for (int i = 0; i < n; ++i) ld += log(u[(size_t)i * n + i]);
    if (logdet) *logdet = 2.0 * ld;
    // Ui = U^-1 (upper), then A^-1 = Ui Ui^T
    std::vector<double> ui((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        ui[(size_t)c * n + c] = 1.0 / u[(size_t)c * n + c];
        for (int i = c - 1; i >= 0; --i) {
            double s = 0.0;
            for (int k = i + 1; k <= c; ++k) s += u[(size_t)i * n + k] * ui[(size_t)k * n + c];
            ui[(size_t)i * n + c] = -s / u[(size_t)i * n + i];
        }
    }
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = 0.0;
            for (int k = j; k < n; ++k) s += ui[(size_t)i * n + k] * ui[(size_t)j * n + k];
            inv[(size_t)i * n + j] = inv[(size_t)j * n + i] = s;
        }
    return true;
}

// cyclic Jacobi for a symmetric matrix (host): eigenvalues descending, vect[k*rank + j] = component k of vector j
void host_sym_eigen(int n, const std::vector<double> &A, int rank, std::vector<double> &vect, std::vector<double> &val)
{
    std::vector<double> a(A), v((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) {
            dg += a[(size_t)i * n + i] * a[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        }
        if (off <= 1e-30 * (dg + off)) break;
        for (int p = 0; p + 1 < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double th = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) { const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q]; a[(size_t)k * n + p] = cs * x - sn * y; a[(size_t)k * n + q] = sn * x + cs * y; }
                for (int k = 0; k < n; ++k) { const double x = a[(size_t)p * n + k], y = a[(size_t)q * n + k]; a[(size_t)p * n + k] = cs * x - sn * y; a[(size_t)q * n + k] = sn * x + cs * y; }
                for (int k = 0; k < n; ++k) { const double x = v[(size_t)k * n + p], y = v[(size_t)k * n + q]; v[(size_t)k * n + p] = cs * x - sn * y; v[(size_t)k * n + q] = sn * x + cs * y; }
            }
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    for (int i = 1; i < n; ++i) { // stable insertion sort, descending
        const int o = ord[i];
        int j = i - 1;
        while (j >= 0 && a[(size_t)ord[j] * n + ord[j]] < a[(size_t)o * n + o]) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = o;
    }
    vect.assign((size_t)n * rank, 0.0);
    val.assign(rank, 0.0);
    for (int j = 0; j < rank; ++j) {
        val[j] = a[(size_t)ord[j] * n + ord[j]];
        for (int k = 0; k < n; ++k) vect[(size_t)k * rank + j] = v[(size_t)k * n + ord[j]];
    }
}

// host copy of a host-or-device array / store of a host vector into a host-or-device array
int fetch_host(gmmiv_ctx *c, const double *p, size_t n, std::vector<double> &out)
{
    out.resize(n);
    if (n == 0 || !p) return GMMIV_OK;
    if (gmmiv_is_device_ptr(p)) {
        GCHK(hipMemcpyAsync(out.data(), p, n * 8, hipMemcpyDeviceToHost, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
    } else memcpy(out.data(), p, n * 8);
    return GMMIV_OK;
}
int store_out(gmmiv_ctx *c, double *p, const std::vector<double> &v)
{
    if (!p) return GMMIV_OK;
    if (gmmiv_is_device_ptr(p)) {
        GCHK(hipMemcpyAsync(p, v.data(), v.size() * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
    } else memcpy(p, v.data(), v.size() * 8);
    return GMMIV_OK;
}

int check_status(gmmiv_ctx *c, int *dstatus, int nb, const char *what)
{
    std::vector<int> h(nb);
    GCHK(hipMemcpyAsync(h.data(), dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    GCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nb; ++i)
        if (h[i]) { gmmiv_set_error("%s: matrix %d of the batch is not positive definite", what, i); return GMMIV_ERR_NUMERIC; }
    return GMMIV_OK;
}

// Scratch set of the batched SPD inverse for nb matrices of order n
struct InvWs {
    double *full = nullptr, *inv = nullptr, *X = nullptr, *invd = nullptr, *panel = nullptr;
    int *status = nullptr;
    int init(gmmiv_ctx *c, int n, int nb)
    {
        const size_t nn = (size_t)n * n;
        const int nblk = (n + 31) / 32;
        void *p;
        int rc;
        if ((rc = c->scratch(WS_T4, nb * nn * 8, &p))) return rc; full = (double *)p;
        if ((rc = c->scratch(WS_T5, nb * nn * 8, &p))) return rc; inv = (double *)p;
        if ((rc = c->scratch(WS_T6, nb * nn * 8, &p))) return rc; X = (double *)p;
        if ((rc = c->scratch(WS_T7, (size_t)nb * nblk * 1024 * 8, &p))) return rc; invd = (double *)p;
        if ((rc = c->scratch(WS_T8, (size_t)nb * n * 32 * 8, &p))) return rc; panel = (double *)p;
        if ((rc = c->scratch(WS_SMALL, (size_t)nb * sizeof(int) + 64, &p))) return rc; status = (int *)p;
        return GMMIV_OK;
    }
};

} // namespace

extern "C" {

size_t gmmiv_tv_packed_len(int R) { return (size_t)R * (R + 1) / 2; }

int gmmiv_tv_subtract_m(gmmiv_ctx *c, int64_t U, int C, int D, const double *N, double *F, const double *means)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || !N || !F || !means) { gmmiv_set_error("tv_subtract_m: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_m;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C))) return rc;
    if ((rc = i_m.init(c, WS_T1, means, SV))) return rc;
    if ((rc = o_f.init(c, WS_T2, F, (size_t)U * SV, true))) return rc;
    c->t_begin("k_subtract_m");
    GCHK(tvk_subtract_m(c->stream, U, C, D, i_n.d, o_f.d, i_m.d));
    c->t_end();
    return o_f.finish();
}

// F_dst = F_src - N ubm_means: restoreStats + substractM of a TotalVariability iteration in one pass over the statistics
int gmmiv_tv_subtract_m_to(gmmiv_ctx *c, int64_t U, int C, int D, const double *N, const double *F_src, double *F_dst, const double *means)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || !N || !F_src || !F_dst || !means) { gmmiv_set_error("tv_subtract_m_to: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_m, i_f;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C))) return rc;
    if ((rc = i_m.init(c, WS_T1, means, SV))) return rc;
    if ((rc = i_f.init(c, WS_T3, F_src, (size_t)U * SV))) return rc;
    if ((rc = o_f.init(c, WS_T2, F_dst, (size_t)U * SV, false))) return rc;
    c->t_begin("k_subtract_m");
    int krc = tvk_subtract_m_to(c->stream, U, C, D, i_n.d, i_f.d, o_f.d, i_m.d);
    if (krc < 0) { // odd vectSize or unaligned rows: copy, then the in-place kernel
        if (o_f.d != i_f.d) GCHK(hipMemcpyAsync(o_f.d, i_f.d, (size_t)U * SV * 8, hipMemcpyDeviceToDevice, c->stream));
        krc = tvk_subtract_m(c->stream, U, C, D, i_n.d, o_f.d, i_m.d);
    }
    GCHK(krc);
    c->t_end();
    return o_f.finish();
}

int gmmiv_tv_tett(gmmiv_ctx *c, int C, int D, int R, const double *Tm, const double *invvar, double *tett_packed)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !Tm || !invvar || !tett_packed) { gmmiv_set_error("tv_tett: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D, P = gmmiv_tv_packed_len(R), RR = (size_t)R * R;
    DevIn<double> i_t, i_iv;
    DevOut<double> o;
    int rc;
    if ((rc = i_t.init(c, WS_T0, Tm, (size_t)R * SV))) return rc;
    if ((rc = i_iv.init(c, WS_T1, invvar, SV))) return rc;
    if ((rc = o.init(c, WS_T2, tett_packed, (size_t)C * P, false))) return rc;
    if (c->tv_tett_direct) { // one kernel: lower triangle only, written packed (tv_kernels.hip: k_tett_packed); D <= 64
        c->t_begin("k_tett_packed");
        const int krc = tvk_tett_packed(c->stream, C, D, R, i_t.d, i_iv.d, o.d);
        c->t_end();
        if (krc == 0) return o.finish();
        if (krc != -1) GCHK(krc);
    }
    void *p;
    if ((rc = c->scratch(WS_T3, (size_t)R * SV * 8, &p))) return rc;
    double *Tiv = (double *)p;
    GCHK(tvk_scale_cols(c->stream, R, (long)SV, i_t.d, i_iv.d, Tiv));
    const int CH = 128;
    if ((rc = c->scratch(WS_T4, (size_t)CH * RR * 8, &p))) return rc;
    double *G = (double *)p;
    c->t_begin("k_dgemm(tett)");
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int nb = (C - c0) < CH ? (C - c0) : CH;
        GCHK(tvk_dgemm(c->stream, false, true, R, R, D, 1.0, i_t.d + (size_t)c0 * D, (long)SV, D, Tiv + (size_t)c0 * D, (long)SV, D,
                       0.0, G, R, (long)RR, nb));
        GCHK(tvk_pack_sym(c->stream, R, nb, G, (long)RR, nullptr, o.d + (size_t)c0 * P, (long)P));
    }
    c->t_end();
    return o.finish();
}

// shared body of estimateW / estimateAandC
static int tv_estep(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                    const double *invvar, const double *tett, double *W, double *A_packed, double *Cmx, double *Rm,
                    double *r, double *meanW, bool accumulate)
{
    GBIND(c);
    const size_t SV = (size_t)C * D, P = gmmiv_tv_packed_len(R), RR = (size_t)R * R;
    int rc;
    DevIn<double> i_n, i_f, i_t, i_iv, i_te;
    DevOut<double> o_w;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C))) return rc;
    if ((rc = i_f.init(c, WS_T1, F, (size_t)U * SV))) return rc;
    if ((rc = i_t.init(c, WS_T2, Tm, (size_t)R * SV))) return rc;
    if ((rc = i_iv.init(c, WS_PART, invvar, SV))) return rc;
    if ((rc = i_te.init(c, WS_X, tett, (size_t)C * P))) return rc;
    if ((rc = o_w.init(c, WS_LSE, W, (size_t)U * R, false))) return rc;
    // accumulators: keep device-side copies when the caller passed host arrays
    void *p;
    double *d_a = nullptr, *d_c = nullptr, *d_rm = nullptr, *d_r = nullptr, *d_mw = nullptr, *d_rp = nullptr;
    // device copies of HOST accumulators live for this call only; the guard frees them on EVERY return path
    struct Owned {
        gmmiv_ctx *c;
        std::vector<void *> v;
        ~Owned() { release(); }
        void release() { if (v.empty()) return; (void)hipStreamSynchronize(c->stream); for (void *q : v) (void)hipFree(q); v.clear(); }
    } owned{c, {}};
    auto dev_acc = [&](double *user, size_t n, double **dev) -> int {
        if (gmmiv_is_device_ptr(user)) { *dev = user; return GMMIV_OK; }
        GCHK(hipMalloc(&p, n * 8));
        owned.v.push_back(p);
        GCHK(hipMemcpyAsync(p, user, n * 8, hipMemcpyHostToDevice, c->stream));
        *dev = (double *)p;
        return GMMIV_OK;
    };
    auto free_owned = [&]() { owned.release(); };
    if (accumulate) {
        if ((rc = dev_acc(A_packed, (size_t)C * P, &d_a)) || (rc = dev_acc(Cmx, (size_t)R * SV, &d_c)) ||
            (rc = dev_acc(Rm, RR, &d_rm)) || (rc = dev_acc(r, R, &d_r)) || (rc = dev_acc(meanW, R, &d_mw))) { free_owned(); return rc; }
        GCHK(hipMalloc(&p, P * 8));
        owned.v.push_back(p);
        d_rp = (double *)p;
        GCHK(hipMemsetAsync(d_rp, 0, P * 8, c->stream));
    }
    if ((rc = c->scratch(WS_TIV, (size_t)R * SV * 8, &p))) { free_owned(); return rc; }
    double *Tiv = (double *)p;
    GCHK(tvk_scale_cols(c->stream, R, (long)SV, i_t.d, i_iv.d, Tiv));

    int BC = (int)c->tv_batch;
    if (BC < 1) BC = 256;
    if (U < BC) BC = (int)U;
    // T-matrix EM: the E_u of a SUPER-BATCH of utterances stay in HBM (tv_acc_mb, default 8 GiB = 13 k utterances at rank 400) and
    // A += N^T E, Cmx += W^T F run ONCE per super-batch with K = its utterance count, instead of once per tv_batch with the 1.3 GB
    // accumulator read and written back every time (config 4's 6250 utterances per rank: 7 x 5.1 ms -> 30 ms for A alone).
    int64_t SB = BC;
    if (accumulate) {
        const int64_t fit = ((int64_t)(c->tv_acc_mb > 0 ? c->tv_acc_mb : 0) << 20) / (int64_t)(P * 8);
        SB = fit / BC * BC;
        if (SB < BC) SB = BC;
        if (SB > U) SB = (U + BC - 1) / BC * BC;
    }
    if ((rc = c->scratch(WS_LP, (size_t)SB * P * 8, &p))) { free_owned(); return rc; }
    double *Lp0 = (double *)p;
    if ((rc = c->scratch(WS_AUX, (size_t)BC * R * 8, &p))) { free_owned(); return rc; }
    double *aux = (double *)p;
    const int nz = tvk_splitk_count(BC, R, (int)SV, c->n_cu);
    size_t slab_doubles = (size_t)nz * BC * R;
    if (accumulate && slab_doubles < (size_t)TVK_BATCH_SUM_SLABS * P) slab_doubles = (size_t)TVK_BATCH_SUM_SLABS * P; // also the partial sums of sum_u E_u
    if (accumulate && slab_doubles < TVK_NARROW_SLABS_DOUBLES(R)) slab_doubles = TVK_NARROW_SLABS_DOUBLES(R); // ... and of sum_u w_u
    if ((rc = c->scratch(WS_SLAB, slab_doubles * 8, &p))) { free_owned(); return rc; }
    double *slabs = (double *)p;
    InvWs ws;
    if ((rc = ws.init(c, R, BC))) { free_owned(); return rc; }

    for (int64_t s0 = 0; s0 < U; s0 += SB) {
    const int64_t ns = (U - s0) < SB ? (U - s0) : SB; // utterances of this super-batch
    for (int64_t u0 = s0; u0 < s0 + ns; u0 += BC) {
        const int nb = (int)((s0 + ns - u0) < BC ? (s0 + ns - u0) : BC);
        const double *Nc = i_n.d + (size_t)u0 * C;
        const double *Fc = i_f.d + (size_t)u0 * SV;
        double *Wc = o_w.d + (size_t)u0 * R;
        double *Lp = Lp0 + (size_t)(u0 - s0) * P;
        GCHK(hipMemsetAsync(ws.status, 0, nb * sizeof(int), c->stream));
        // L (packed) = N * TETt ; + I on unpack
        c->t_begin("k_dgemm(L)");
        GCHK(tvk_dgemm(c->stream, false, false, nb, (int)P, C, 1.0, Nc, C, 0, i_te.d, (long)P, 0, 0.0, Lp, (long)P, 0, 1));
        c->t_end();
        const bool packed_in = tvk_chol_accepts_packed(R); // the factorisation reads the packed GEMM result (+ I) itself
        if (!packed_in) GCHK(tvk_unpack_sym(c->stream, R, nb, Lp, (long)P, ws.full, 1.0));
        // aux = F Sigma^-1 T^T
        GCHK(tvk_dgemm_splitk(c->stream, false, true, nb, R, (int)SV, 1.0, Fc, (long)SV, Tiv, (long)SV, 0.0, aux, R, nz, slabs));
        if (accumulate) { // the T-matrix EM needs L^-1 itself (E = L^-1 + w w^T): explicit inverse like the reference
            if (packed_in) { // w by substitution, E = L^-1 + w w^T straight into the packed buffer
                GCHK(tvk_inverse_e_packed_batched(c->stream, R, nb, ws.full, ws.X, ws.invd, ws.status, Lp, (long)P, 1.0, aux, Wc));
            } else {
                GCHK(tvk_spd_inverse_batched(c->stream, R, nb, ws.full, ws.inv, ws.X, ws.invd, ws.panel, ws.status));
                GCHK(tvk_batched_matvec(c->stream, R, nb, ws.inv, aux, Wc));
            }
        } else {          // extraction only needs w = L^-1 aux: Cholesky + two triangular solves
            if (packed_in) GCHK(tvk_chol_left_batched(c->stream, R, nb, ws.full, ws.invd, ws.status, Lp, (long)P, 1.0));
            else GCHK(tvk_chol_batched(c->stream, R, nb, ws.full, ws.invd, ws.panel, ws.status));
            GCHK(tvk_chol_solve_batched(c->stream, R, nb, ws.full, ws.invd, aux, Wc));
        }
        if ((rc = check_status(c, ws.status, nb, "tv: L"))) { free_owned(); return rc; }
        // E = L^-1 + w w^T (packed, in the super-batch buffer)
        if (accumulate && !packed_in) GCHK(tvk_pack_sym(c->stream, R, nb, ws.inv, (long)RR, Wc, Lp, (long)P));
    }
    if (accumulate) {
        // A += N^T E ; Cmx += W^T F ; R += sum E ; r, meanW += sum w    over the ns utterances of the super-batch
        const double *Ns = i_n.d + (size_t)s0 * C, *Fs = i_f.d + (size_t)s0 * SV, *Ws = o_w.d + (size_t)s0 * R;
        GCHK(tvk_dgemm(c->stream, true, false, C, (int)P, (int)ns, 1.0, Ns, C, 0, Lp0, (long)P, 0, 1.0, d_a, (long)P, 0, 1));
        // A is complete once the last super-batch's GEMM is enqueued: a caller that shards the M-step starts its exchange here,
        // under the Cmx GEMM and the batch sums below (gmmiv_ctx_set_hook "tv_a_ready"; device accumulators only)
        if (s0 + SB >= U && d_a == A_packed) { c->hook_tv_a_ready.call(); GBIND(c); } // the hook may have driven another context on this thread: bind ours again
        GCHK(tvk_dgemm(c->stream, true, false, R, (int)SV, (int)ns, 1.0, Ws, R, 0, Fs, (long)SV, 0, 1.0, d_c, (long)SV, 0, 1));
        GCHK(tvk_batch_sum(c->stream, (long)P, (int)ns, Lp0, (long)P, d_rp, slabs)); // slabs (split-K workspace of aux) is free again
        GCHK(tvk_colsum_narrow(c->stream, R, (int)ns, Ws, R, d_r, d_mw, slabs)); // r and meanW both accumulate sum_u w_u; slabs is free again (stream order)
    }
    }
    if (accumulate) {
        GCHK(tvk_add_unpacked(c->stream, R, d_rp, d_rm));
        auto back = [&](double *user, double *dev, size_t n) -> int {
            if (dev != user) GCHK(hipMemcpyAsync(user, dev, n * 8, hipMemcpyDeviceToHost, c->stream));
            return GMMIV_OK;
        };
        if ((rc = back(A_packed, d_a, (size_t)C * P)) || (rc = back(Cmx, d_c, (size_t)R * SV)) || (rc = back(Rm, d_rm, RR)) ||
            (rc = back(r, d_r, R)) || (rc = back(meanW, d_mw, R))) { free_owned(); return rc; }
    }
    rc = o_w.finish();
    free_owned();
    return rc;
}

int gmmiv_tv_estimate_w(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                        const double *invvar, const double *tett_packed, double *W)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || R <= 0 || !N || !F || !Tm || !invvar || !tett_packed || !W) { gmmiv_set_error("tv_estimate_w: bad argument"); return GMMIV_ERR_ARG; }
    if (U == 0) return GMMIV_OK;
    return tv_estep(c, U, C, D, R, N, F, Tm, invvar, tett_packed, W, nullptr, nullptr, nullptr, nullptr, nullptr, false);
}

int gmmiv_tv_estimate_a_and_c(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, const double *F,
                              const double *Tm, const double *invvar, const double *tett_packed, double *W,
                              double *A_packed, double *Cmx, double *Rm, double *r, double *meanW)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || R <= 0 || !N || !F || !Tm || !invvar || !tett_packed || !W || !A_packed || !Cmx || !Rm || !r || !meanW) { gmmiv_set_error("tv_estimate_a_and_c: bad argument"); return GMMIV_ERR_ARG; }
    if (U == 0) return GMMIV_OK;
    return tv_estep(c, U, C, D, R, N, F, Tm, invvar, tett_packed, W, A_packed, Cmx, Rm, r, meanW, true);
}

int gmmiv_tv_update_t(gmmiv_ctx *c, int C, int D, int R, const double *A_packed, const double *Cmx, double *Tm)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !A_packed || !Cmx || !Tm) { gmmiv_set_error("tv_update_t: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D, P = gmmiv_tv_packed_len(R), RR = (size_t)R * R;
    int rc;
    DevIn<double> i_a, i_c;
    DevOut<double> o_t;
    if ((rc = i_a.init(c, WS_T0, A_packed, (size_t)C * P))) return rc;
    if ((rc = i_c.init(c, WS_T1, Cmx, (size_t)R * SV))) return rc;
    if ((rc = o_t.init(c, WS_T2, Tm, (size_t)R * SV, false))) return rc;
    int CH = c->n_cu > 128 ? c->n_cu : 128; // one workgroup per matrix in the batched inverse: a batch fills the chip
    if (C < CH) CH = C;
    InvWs ws;
    if ((rc = ws.init(c, R, CH))) return rc;
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int nb = (C - c0) < CH ? (C - c0) : CH;
        GCHK(hipMemsetAsync(ws.status, 0, nb * sizeof(int), c->stream));
        if (tvk_chol_accepts_packed(R) && D <= 64 && c->tv_mstep_solve) {
            // T_c = A_c^-1 Cmx_c by substitution through the Cholesky factor: 60 right-hand sides per Gaussian, no explicit inverse
            // (the reference inverts, :981-1000 -- same result to rounding, a third of the work)
            GCHK(tvk_chol_left_batched(c->stream, R, nb, ws.full, ws.invd, ws.status, i_a.d + (size_t)c0 * P, (long)P, 0.0));
            GCHK(tvk_chol_solve_multi_batched(c->stream, R, nb, D, ws.full, ws.invd, i_c.d + (size_t)c0 * D, (long)SV, D, o_t.d + (size_t)c0 * D,
                                              (long)SV, D));
        } else {
            if (tvk_chol_accepts_packed(R)) {
                GCHK(tvk_spd_inverse_left_batched(c->stream, R, nb, ws.full, ws.inv, ws.X, ws.invd, ws.status, i_a.d + (size_t)c0 * P, (long)P, 0.0));
            } else {
                GCHK(tvk_unpack_sym(c->stream, R, nb, i_a.d + (size_t)c0 * P, (long)P, ws.full, 0.0));
                GCHK(tvk_spd_inverse_batched(c->stream, R, nb, ws.full, ws.inv, ws.X, ws.invd, ws.panel, ws.status));
            }
            // T_c = A_c^-1 Cmx_c
            GCHK(tvk_dgemm(c->stream, false, false, R, D, R, 1.0, ws.inv, R, (long)RR, i_c.d + (size_t)c0 * D, (long)SV, D, 0.0,
                           o_t.d + (size_t)c0 * D, (long)SV, D, nb));
        }
        if ((rc = check_status(c, ws.status, nb, "tv_update_t: A_c"))) return rc;
    }
    return o_t.finish();
}

int gmmiv_tv_min_divergence(gmmiv_ctx *c, int C, int D, int R, double n_sessions, double *Rm, double *r,
                            const double *meanW, double *ubm_means, double *Tm)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !(n_sessions > 0) || !Rm || !r || !meanW || !ubm_means || !Tm) { gmmiv_set_error("tv_min_divergence: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D, RR = (size_t)R * R;
    int rc;
    DevOut<double> o_rm, o_r, o_mean, o_t;
    DevIn<double> i_mw;
    if ((rc = o_rm.init(c, WS_T0, Rm, RR, true))) return rc;
    if ((rc = o_r.init(c, WS_T1, r, R, true))) return rc;
    if ((rc = i_mw.init(c, WS_T2, meanW, R))) return rc;
    if ((rc = o_mean.init(c, WS_T3, ubm_means, SV, true))) return rc;
    if ((rc = o_t.init(c, WS_T4, Tm, (size_t)R * SV, true))) return rc;
    void *p;
    if ((rc = c->scratch(WS_T5, RR * 8, &p))) return rc;
    double *dCh = (double *)p;
    if (tvk_chol_accepts_packed(R) && c->tv_md_device) {
        // R <- R / n - r r^T and its factor on the device: one workgroup of k_chol_left (R = L L^T, Ch = L^T); the host only sees the
        // status word.  (The host route below cost 4-5 ms of a 130 ms iteration at R = 400: two 1.28 MB copies each way and a scalar
        // O(R^3) loop.)
        if ((rc = c->scratch(WS_T8, RR * 8, &p))) return rc;
        double *work = (double *)p;
        if ((rc = c->scratch(WS_T7, (size_t)((R + 31) / 32) * 1024 * 8, &p))) return rc;
        double *invd = (double *)p;
        if ((rc = c->scratch(WS_SMALL, sizeof(int) + 64, &p))) return rc;
        int *status = (int *)p;
        GCHK(hipMemsetAsync(status, 0, sizeof(int), c->stream));
        GCHK(tvk_md_normalize(c->stream, R, n_sessions, o_rm.d, o_r.d, work));
        GCHK(tvk_chol_left_batched(c->stream, R, 1, work, invd, status));
        GCHK(tvk_lower_to_upper(c->stream, R, work, dCh));
        if (check_status(c, status, 1, "tv_min_divergence: R")) { gmmiv_set_error("tv_min_divergence: R is not positive definite"); return GMMIV_ERR_NUMERIC; }
    } else {
        // R x R normalisation + Cholesky on the host (odd R: the device factorisation wants 16-byte rows)
        std::vector<double> hR(RR), hr(R), ch;
        GCHK(hipMemcpyAsync(hR.data(), o_rm.d, RR * 8, hipMemcpyDeviceToHost, c->stream));
        GCHK(hipMemcpyAsync(hr.data(), o_r.d, R * 8, hipMemcpyDeviceToHost, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < R; ++i) hr[i] /= n_sessions;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j) hR[(size_t)i * R + j] = hR[(size_t)i * R + j] / n_sessions - hr[i] * hr[j];
        if (!host_cholesky_upper(R, hR, ch)) { gmmiv_set_error("tv_min_divergence: R is not positive definite"); return GMMIV_ERR_NUMERIC; }
        GCHK(hipMemcpyAsync(o_rm.d, hR.data(), RR * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipMemcpyAsync(o_r.d, hr.data(), R * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipMemcpyAsync(dCh, ch.data(), RR * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipStreamSynchronize(c->stream)); // the host vectors go out of scope
    }
    // R is factored, T has not been read yet: a caller whose T is still arriving (all-gather begun before the call) joins it here
    if (o_t.d == Tm) { c->hook_md_factored.call(); GBIND(c); } // (see tv_a_ready)
    // mean += T^T meanW (old T), then T <- Ch T
    GCHK(tvk_vecmat_add(c->stream, R, (long)SV, i_mw.d, o_t.d, o_mean.d));
    if ((rc = c->scratch(WS_T6, (size_t)R * SV * 8, &p))) return rc;
    double *Tn = (double *)p;
    GCHK(tvk_dgemm(c->stream, false, false, R, (int)SV, R, 1.0, dCh, R, 0, o_t.d, (long)SV, 0, 0.0, Tn, (long)SV, 0, 1));
    GCHK(hipMemcpyAsync(o_t.d, Tn, (size_t)R * SV * 8, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = o_rm.finish()) || (rc = o_r.finish()) || (rc = o_mean.finish())) return rc;
    return o_t.finish();
}

// ---- approximate extractors ------------------------------------------------------------------
int gmmiv_tv_norm_statistics(gmmiv_ctx *c, int64_t U, int C, int D, const double *N, double *F, const double *means,
                             const double *invvar)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || !N || !F || !means || !invvar) { gmmiv_set_error("tv_norm_statistics: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_m, i_v;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C)) || (rc = i_m.init(c, WS_T2, means, SV)) || (rc = i_v.init(c, WS_T3, invvar, SV)) ||
        (rc = o_f.init(c, WS_T1, F, (size_t)U * SV, true))) return rc;
    GCHK(tvk_norm_stats(c->stream, (long)U, C, D, i_n.d, o_f.d, i_m.d, i_v.d));
    return o_f.finish();
}

int gmmiv_tv_subtract_m_plus_tw(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, double *F, const double *means,
                                const double *Tm, const double *W)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || R <= 0 || !N || !F || !means || !Tm || !W) { gmmiv_set_error("tv_subtract_m_plus_tw: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_m, i_t, i_w;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C)) || (rc = i_m.init(c, WS_T2, means, SV)) || (rc = i_t.init(c, WS_T3, Tm, (size_t)R * SV)) ||
        (rc = i_w.init(c, WS_LSE, W, (size_t)U * R)) || (rc = o_f.init(c, WS_T1, F, (size_t)U * SV, true))) return rc;
    const int tvb = c->tv_batch > 0 ? (int)c->tv_batch : 256;
    const int BC = U < tvb ? (int)(U > 0 ? U : 1) : tvb;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)BC * SV * 8, &p))) return rc;
    double *TW = (double *)p;
    for (int64_t u0 = 0; u0 < U; u0 += BC) {
        const int nb = (int)((U - u0) < BC ? (U - u0) : BC);
        GCHK(tvk_dgemm(c->stream, false, false, nb, (int)SV, R, 1.0, i_w.d + (size_t)u0 * R, R, 0, i_t.d, (long)SV, 0, 0.0, TW, (long)SV, 0, 1));
        GCHK(tvk_sub_mtw(c->stream, nb, C, D, i_n.d + (size_t)u0 * C, o_f.d + (size_t)u0 * SV, i_m.d, TW));
    }
    return o_f.finish();
}

// ---- JFA (LIA_SpkTools/src/AccumulateJFAStat.cpp).  The factor steps themselves are the TV functions above under other
// names: estimateVEVT / estimateUEUT = gmmiv_tv_tett, estimateAndInverseL_E{V,C} + estimate{YandV,XandU} =
// gmmiv_tv_estimate_a_and_c, estimateY / estimateX = gmmiv_tv_estimate_w, update{V,U}estimate = gmmiv_tv_update_t. ----
int gmmiv_jfa_subtract(gmmiv_ctx *c, int64_t rows, int C, int D, const double *N, double *F, const int64_t *owner, int64_t nfact,
                       const double *means, int R, const double *Tm, const double *W, const double *Dm, const double *Z)
{
    if (!c || rows < 0 || C <= 0 || D <= 0 || !N || !F || nfact < 0 || (Tm && (R <= 0 || !W)) || (Dm && !Z)) { gmmiv_set_error("jfa_subtract: bad argument"); return GMMIV_ERR_ARG; }
    if (rows == 0) return GMMIV_OK;
    GBIND(c);
    const size_t SV = (size_t)C * D;
    if (!owner && nfact < rows && (Tm || Dm)) { gmmiv_set_error("jfa_subtract: %lld factor rows for %lld statistics rows and no owner map", (long long)nfact, (long long)rows); return GMMIV_ERR_ARG; }
    if (owner && !gmmiv_is_device_ptr(owner))
        for (int64_t r = 0; r < rows; ++r)
            if (owner[r] < 0 || owner[r] >= nfact) { gmmiv_set_error("jfa_subtract: owner[%lld] = %lld out of range", (long long)r, (long long)owner[r]); return GMMIV_ERR_ARG; }
    DevIn<double> i_n, i_m, i_t, i_w, i_d, i_z;
    DevIn<int64_t> i_o;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)rows * C)) || (rc = i_m.init(c, WS_T2, means, SV)) || (rc = i_t.init(c, WS_T3, Tm, Tm ? (size_t)R * SV : 0)) ||
        (rc = i_w.init(c, WS_LSE, Tm ? W : nullptr, Tm ? (size_t)nfact * R : 0)) || (rc = i_d.init(c, WS_T4, Dm, SV)) ||
        (rc = i_z.init(c, WS_T5, Dm ? Z : nullptr, Dm ? (size_t)nfact * SV : 0)) || (rc = i_o.init(c, WS_SEG, owner, (size_t)rows)) ||
        (rc = o_f.init(c, WS_T1, F, (size_t)rows * SV, true))) return rc;
    const int tvb = c->tv_batch > 0 ? (int)c->tv_batch : 256;
    const int BC = rows < tvb ? (int)rows : tvb;
    double *TW = nullptr, *Wg = nullptr;
    void *p;
    if (Tm) {
        if ((rc = c->scratch(WS_TIV, (size_t)BC * SV * 8, &p))) return rc;
        TW = (double *)p;
        if ((rc = c->scratch(WS_AUX, (size_t)BC * R * 8, &p))) return rc;
        Wg = (double *)p;
    }
    for (int64_t r0 = 0; r0 < rows; r0 += BC) {
        const int nb = (int)((rows - r0) < BC ? (rows - r0) : BC);
        if (Tm) {
            GCHK(tvk_gather_rows(c->stream, nb, R, (long)r0, (const long *)i_o.d, i_w.d, Wg));
            GCHK(tvk_dgemm(c->stream, false, false, nb, (int)SV, R, 1.0, Wg, R, 0, i_t.d, (long)SV, 0, 0.0, TW, (long)SV, 0, 1));
        }
        GCHK(tvk_jfa_sub(c->stream, nb, C, D, (long)r0, (const long *)i_o.d, i_n.d, o_f.d, i_m.d, TW, i_d.d, i_z.d));
    }
    return o_f.finish();
}

int gmmiv_jfa_subtract_sessions(gmmiv_ctx *c, int64_t nspk, const int64_t *sess_begin, int C, int D, const double *N_h, double *F_X,
                                int R, const double *Um, const double *X)
{
    if (!c || nspk < 0 || !sess_begin || C <= 0 || D <= 0 || R <= 0 || !N_h || !F_X || !Um || !X) { gmmiv_set_error("jfa_subtract_sessions: bad argument"); return GMMIV_ERR_ARG; }
    if (gmmiv_is_device_ptr(sess_begin)) { gmmiv_set_error("jfa_subtract_sessions: sess_begin must be a host array"); return GMMIV_ERR_ARG; }
    if (nspk == 0) return GMMIV_OK;
    for (int64_t s = 0; s < nspk; ++s)
        if (sess_begin[s + 1] < sess_begin[s] || sess_begin[0] != 0) { gmmiv_set_error("jfa_subtract_sessions: sess_begin must start at 0 and be non-decreasing"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    const int64_t nsess = sess_begin[nspk];
    if (nsess == 0) return GMMIV_OK;
    DevIn<double> i_n, i_u, i_x;
    DevIn<int64_t> i_b;
    DevOut<double> o_f;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N_h, (size_t)nsess * C)) || (rc = i_u.init(c, WS_T3, Um, (size_t)R * SV)) || (rc = i_x.init(c, WS_LSE, X, (size_t)nsess * R)) ||
        (rc = i_b.init(c, WS_SEG, sess_begin, (size_t)nspk + 1)) || (rc = o_f.init(c, WS_T1, F_X, (size_t)nspk * SV, true))) return rc;
    const int tvb = c->tv_batch > 0 ? (int)c->tv_batch : 256;
    const int BC = nsess < tvb ? (int)nsess : tvb;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)BC * SV * 8, &p))) return rc;
    double *G = (double *)p;
    int64_t s_lo = 0;
    for (int64_t h0 = 0; h0 < nsess; h0 += BC) {
        const int64_t h1 = (h0 + BC) < nsess ? (h0 + BC) : nsess;
        GCHK(tvk_dgemm(c->stream, false, false, (int)(h1 - h0), (int)SV, R, 1.0, i_x.d + (size_t)h0 * R, R, 0, i_u.d, (long)SV, 0, 0.0, G, (long)SV, 0, 1));
        while (s_lo < nspk && sess_begin[s_lo + 1] <= h0) ++s_lo;      // first speaker with a session in [h0, h1)
        int64_t s_hi = s_lo;
        while (s_hi < nspk && sess_begin[s_hi] < h1) ++s_hi;           // one past the last
        GCHK(tvk_jfa_sub_sessions(c->stream, (long)s_lo, (long)(s_hi - s_lo), (long)h0, (long)h1, C, D, (const long *)i_b.d, i_n.d, G, o_f.d));
    }
    return o_f.finish();
}

int gmmiv_jfa_estimate_z(gmmiv_ctx *c, int64_t nspk, int C, int D, const double *N, const double *F, const double *invvar, const double *Dm,
                         double tau, double *Z)
{
    if (!c || nspk < 0 || C <= 0 || D <= 0 || !N || !F || !invvar || !Dm || !Z) { gmmiv_set_error("jfa_estimate_z: bad argument"); return GMMIV_ERR_ARG; }
    if (nspk == 0) return GMMIV_OK;
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_f, i_v, i_d;
    DevOut<double> o_z;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)nspk * C)) || (rc = i_f.init(c, WS_T1, F, (size_t)nspk * SV)) || (rc = i_v.init(c, WS_T2, invvar, SV)) ||
        (rc = i_d.init(c, WS_T4, Dm, SV)) || (rc = o_z.init(c, WS_T5, Z, (size_t)nspk * SV, false))) return rc;
    GCHK(tvk_jfa_z(c->stream, (long)nspk, C, D, i_n.d, i_f.d, i_v.d, i_d.d, tau, o_z.d));
    return o_z.finish();
}

int gmmiv_jfa_estimate_z_and_d(gmmiv_ctx *c, int64_t nspk, int C, int D, const double *N, const double *F, const double *invvar, double *Dm,
                               double *Z)
{
    if (!c || nspk <= 0 || C <= 0 || D <= 0 || !N || !F || !invvar || !Dm || !Z) { gmmiv_set_error("jfa_estimate_z_and_d: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_f, i_v;
    DevOut<double> o_d, o_z;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)nspk * C)) || (rc = i_f.init(c, WS_T1, F, (size_t)nspk * SV)) || (rc = i_v.init(c, WS_T2, invvar, SV)) ||
        (rc = o_d.init(c, WS_T4, Dm, SV, true)) || (rc = o_z.init(c, WS_T5, Z, (size_t)nspk * SV, false))) return rc;
    GCHK(tvk_jfa_z_and_d(c->stream, (long)nspk, C, D, i_n.d, i_f.d, i_v.d, o_d.d, o_z.d));
    if ((rc = o_d.finish())) return rc;
    return o_z.finish();
}

int gmmiv_tv_norm_t(gmmiv_ctx *c, int C, int D, int R, double *Tm, const double *invvar)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !Tm || !invvar) { gmmiv_set_error("tv_norm_t: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_v;
    DevOut<double> o_t;
    int rc;
    if ((rc = i_v.init(c, WS_T0, invvar, SV)) || (rc = o_t.init(c, WS_T1, Tm, (size_t)R * SV, true))) return rc;
    GCHK(tvk_scale_cols_fn(c->stream, R, (long)SV, D, 0, o_t.d, i_v.d, o_t.d));
    return o_t.finish();
}

int gmmiv_tv_weighted_cov(gmmiv_ctx *c, int C, int D, int R, const double *Tm, const double *weight, double *Wm)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !Tm || !weight || !Wm) { gmmiv_set_error("tv_weighted_cov: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_t, i_w;
    DevOut<double> o;
    int rc;
    if ((rc = i_t.init(c, WS_T1, Tm, (size_t)R * SV)) || (rc = i_w.init(c, WS_T0, weight, C)) || (rc = o.init(c, WS_T2, Wm, (size_t)R * R, false))) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)R * SV * 8, &p))) return rc;
    double *Ts = (double *)p;
    GCHK(tvk_scale_cols_fn(c->stream, R, (long)SV, D, 1, i_t.d, i_w.d, Ts));
    const int nz = tvk_splitk_count(R, R, (int)SV, c->n_cu);
    if ((rc = c->scratch(WS_SLAB, (size_t)nz * R * R * 8, &p))) return rc;
    GCHK(tvk_dgemm_splitk(c->stream, false, true, R, R, (int)SV, 1.0, Ts, (long)SV, i_t.d, (long)SV, 0.0, o.d, R, nz, (double *)p));
    return o.finish();
}

int gmmiv_tv_approximate_tctc(gmmiv_ctx *c, int C, int D, int R, const double *Tm, const double *Q, double *Dm)
{
    if (!c || C <= 0 || D <= 0 || R <= 0 || !Tm || !Q || !Dm) { gmmiv_set_error("tv_approximate_tctc: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_t, i_q;
    DevOut<double> o;
    int rc;
    if ((rc = i_t.init(c, WS_T1, Tm, (size_t)R * SV)) || (rc = i_q.init(c, WS_T0, Q, (size_t)R * R)) || (rc = o.init(c, WS_T2, Dm, (size_t)C * R, true))) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, SV * R * 8, &p))) return rc;
    double *A = (double *)p; // [SV x R] = T^T Q
    GCHK(tvk_dgemm(c->stream, true, false, (int)SV, R, R, 1.0, i_t.d, (long)SV, 0, i_q.d, R, 0, 0.0, A, R, 0, 1));
    GCHK(tvk_block_colnorm(c->stream, C, D, R, A, o.d));
    return o.finish();
}

// shared front end of the two approximate estimators: aux[nb x R] = F_chunk T^T (T and F normalised, no invvar)
static int approx_aux(gmmiv_ctx *c, int nb, int R, size_t SV, const double *Fc, const double *Td, double *aux, int nz, double *slabs)
{
    GCHK(tvk_dgemm_splitk(c->stream, false, true, nb, R, (int)SV, 1.0, Fc, (long)SV, Td, (long)SV, 0.0, aux, R, nz, slabs));
    return GMMIV_OK;
}

int gmmiv_tv_estimate_w_ubm_weight(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                                   const double *Wm, double *W)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || R <= 0 || !N || !F || !Tm || !Wm || !W) { gmmiv_set_error("tv_estimate_w_ubm_weight: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_f, i_t, i_w;
    DevOut<double> o;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C)) || (rc = i_f.init(c, WS_T1, F, (size_t)U * SV)) || (rc = i_t.init(c, WS_T2, Tm, (size_t)R * SV)) ||
        (rc = i_w.init(c, WS_T3, Wm, (size_t)R * R)) || (rc = o.init(c, WS_LSE, W, (size_t)U * R, true))) return rc;
    const int tvb = c->tv_batch > 0 ? (int)c->tv_batch : 256;
    const int BC = U < tvb ? (int)(U > 0 ? U : 1) : tvb;
    void *p;
    if ((rc = c->scratch(WS_AUX, (size_t)2 * BC * R * 8, &p))) return rc;
    double *aux = (double *)p, *wc = aux + (size_t)BC * R;
    const int nz = tvk_splitk_count(BC, R, (int)SV, c->n_cu);
    if ((rc = c->scratch(WS_SLAB, (size_t)nz * BC * R * 8, &p))) return rc;
    double *slabs = (double *)p;
    InvWs ws;
    if ((rc = ws.init(c, R, BC))) return rc;
    for (int64_t u0 = 0; u0 < U; u0 += BC) {
        const int nb = (int)((U - u0) < BC ? (U - u0) : BC);
        GCHK(hipMemsetAsync(ws.status, 0, nb * sizeof(int), c->stream));
        GCHK(tvk_build_l_ubm(c->stream, R, C, nb, i_n.d + (size_t)u0 * C, i_w.d, ws.full));
        if ((rc = approx_aux(c, nb, R, SV, i_f.d + (size_t)u0 * SV, i_t.d, aux, nz, slabs))) return rc;
        GCHK(tvk_chol_batched(c->stream, R, nb, ws.full, ws.invd, ws.panel, ws.status));
        GCHK(tvk_chol_solve_batched(c->stream, R, nb, ws.full, ws.invd, aux, wc));
        if ((rc = check_status(c, ws.status, nb, "tv_estimate_w_ubm_weight: L"))) return rc;
        GCHK(tvk_axpby(c->stream, (long)nb * R, 1.0, wc, 1.0, o.d + (size_t)u0 * R, o.d + (size_t)u0 * R));
    }
    return o.finish();
}

int gmmiv_tv_estimate_w_eigen(gmmiv_ctx *c, int64_t U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                              const double *Dm, const double *Q, double *W)
{
    if (!c || U < 0 || C <= 0 || D <= 0 || R <= 0 || !N || !F || !Tm || !Dm || !Q || !W) { gmmiv_set_error("tv_estimate_w_eigen: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t SV = (size_t)C * D;
    DevIn<double> i_n, i_f, i_t, i_d, i_q;
    DevOut<double> o;
    int rc;
    if ((rc = i_n.init(c, WS_T0, N, (size_t)U * C)) || (rc = i_f.init(c, WS_T1, F, (size_t)U * SV)) || (rc = i_t.init(c, WS_T2, Tm, (size_t)R * SV)) ||
        (rc = i_d.init(c, WS_T3, Dm, (size_t)C * R)) || (rc = i_q.init(c, WS_T4, Q, (size_t)R * R)) || (rc = o.init(c, WS_LSE, W, (size_t)U * R, true))) return rc;
    const int tvb = c->tv_batch > 0 ? (int)c->tv_batch : 256;
    const int BC = U < tvb ? (int)(U > 0 ? U : 1) : tvb;
    void *p;
    if ((rc = c->scratch(WS_AUX, (size_t)3 * BC * R * 8, &p))) return rc;
    double *aux = (double *)p, *nd = aux + (size_t)BC * R, *b = nd + (size_t)BC * R;
    const int nz = tvk_splitk_count(BC, R, (int)SV, c->n_cu);
    if ((rc = c->scratch(WS_SLAB, (size_t)nz * BC * R * 8, &p))) return rc;
    double *slabs = (double *)p;
    for (int64_t u0 = 0; u0 < U; u0 += BC) {
        const int nb = (int)((U - u0) < BC ? (U - u0) : BC);
        if ((rc = approx_aux(c, nb, R, SV, i_f.d + (size_t)u0 * SV, i_t.d, aux, nz, slabs))) return rc;
        GCHK(tvk_dgemm(c->stream, false, false, nb, R, C, 1.0, i_n.d + (size_t)u0 * C, C, 0, i_d.d, R, 0, 0.0, nd, R, 0, 1));   // N Dm
        GCHK(tvk_dgemm(c->stream, false, false, nb, R, R, 1.0, aux, R, 0, i_q.d, R, 0, 0.0, b, R, 0, 1));                          // (Q^T aux)^T = aux Q
        GCHK(tvk_mul_recip1p(c->stream, (long)nb * R, b, nd));
        GCHK(tvk_dgemm(c->stream, false, true, nb, R, R, 1.0, b, R, 0, i_q.d, R, 0, 1.0, o.d + (size_t)u0 * R, R, 0, 1));         // += b Q^T
    }
    return o.finish();
}

int gmmiv_tv_orthonormalize_t(gmmiv_ctx *c, int R, int64_t SV, double *Tm)
{
    if (!c || R <= 0 || SV <= 0 || !Tm) { gmmiv_set_error("tv_orthonormalize_t: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    DevOut<double> o;
    int rc;
    if ((rc = o.init(c, WS_T0, Tm, (size_t)R * SV, true))) return rc;
    void *p;
    if ((rc = c->scratch(WS_T1, (size_t)R * SV * 8, &p))) return rc;
    double *Q = (double *)p;
    // Gram-Schmidt on the rows of T is T = L Q with L lower triangular, positive diagonal: Q = L^-1 T with L the Cholesky factor
    // of the Gram matrix T T^T -- two GEMMs and an R x R factorisation on the host (3 ms at R = 400, SV = 122 880) instead of R
    // dependent projection steps (116 ms).  Same result as the reference's classical Gram-Schmidt up to cond(T)^2 eps, which is
    // also what the classical scheme itself is good for; zero / dependent rows (no Cholesky factor) and badly conditioned T
    // (diagonal ratio of L below 1e-4) keep the step-by-step kernel, which reproduces the reference's zero-row rule.
    {
        const size_t RR = (size_t)R * R;
        const int nz = tvk_splitk_count(R, R, (int)SV, c->n_cu);
        void *q;
        if ((rc = c->scratch(WS_SLAB, (size_t)(nz > 1 ? nz : 1) * RR * 8, &q))) return rc;
        if ((rc = c->scratch(WS_T3, 2 * RR * 8, &p))) return rc;
        double *dG = (double *)p, *dLi = dG + RR;
        GCHK(tvk_dgemm_splitk(c->stream, false, true, R, R, (int)SV, 1.0, o.d, (long)SV, o.d, (long)SV, 0.0, dG, R, nz, (double *)q));
        std::vector<double> G;
        if ((rc = fetch_host(c, dG, RR, G))) return rc;
        std::vector<double> L(RR, 0.0), Li(RR, 0.0);
        bool ok = true;
        double dmin = __builtin_inf(), dmax = 0.0;
        for (int j = 0; j < R && ok; ++j) { // lower Cholesky, column by column
            double d = G[(size_t)j * R + j];
            for (int k = 0; k < j; ++k) d -= L[(size_t)j * R + k] * L[(size_t)j * R + k];
            if (!(d > 0.0)) { ok = false; break; }
            const double ljj = sqrt(d);
            L[(size_t)j * R + j] = ljj;
            dmin = ljj < dmin ? ljj : dmin;
            dmax = ljj > dmax ? ljj : dmax;
            for (int i = j + 1; i < R; ++i) {
                double t = G[(size_t)i * R + j];
                for (int k = 0; k < j; ++k) t -= L[(size_t)i * R + k] * L[(size_t)j * R + k];
                L[(size_t)i * R + j] = t / ljj;
            }
        }
        if (ok && dmin > 1e-4 * dmax) {
            for (int cidx = 0; cidx < R; ++cidx) { // Li = L^-1 by forward substitution, column by column
                Li[(size_t)cidx * R + cidx] = 1.0 / L[(size_t)cidx * R + cidx];
                for (int i = cidx + 1; i < R; ++i) {
                    double t = 0.0;
                    for (int k = cidx; k < i; ++k) t += L[(size_t)i * R + k] * Li[(size_t)k * R + cidx];
                    Li[(size_t)i * R + cidx] = -t / L[(size_t)i * R + i];
                }
            }
            GCHK(hipMemcpyAsync(dLi, Li.data(), RR * 8, hipMemcpyHostToDevice, c->stream));
            GCHK(tvk_dgemm(c->stream, false, false, R, (int)SV, R, 1.0, dLi, R, 0, o.d, (long)SV, 0, 0.0, Q, (long)SV, 0, 1));
            GCHK(hipMemcpyAsync(o.d, Q, (size_t)R * SV * 8, hipMemcpyDeviceToDevice, c->stream));
            GCHK(hipStreamSynchronize(c->stream)); // Li is a stack-lifetime vector
            return o.finish();
        }
    }
    if ((rc = c->scratch(WS_T2, ((size_t)SV + R + 512) * 8, &p))) return rc;
    double *v = (double *)p, *rv = v + SV, *partial = rv + R;
    GCHK(tvk_orthonormalize(c->stream, R, (long)SV, o.d, Q, rv, v, partial));
    GCHK(hipMemcpyAsync(o.d, Q, (size_t)R * SV * 8, hipMemcpyDeviceToDevice, c->stream));
    return o.finish();
}

// ---- i-vector normalisation ------------------------------------------------------------------
int gmmiv_iv_normalize(gmmiv_ctx *c, int dim_in, int dim_out, int64_t n, const double *X, const double *mean,
                       const double *M, int length_norm, double *Y)
{
    if (!c || dim_in <= 0 || dim_out <= 0 || n < 0 || !X || !Y) { gmmiv_set_error("iv_normalize: bad argument"); return GMMIV_ERR_ARG; }
    if (!M && dim_in != dim_out) { gmmiv_set_error("iv_normalize: dim_out must equal dim_in without a rotation matrix"); return GMMIV_ERR_ARG; }
    if (n > 0x7fffffff) { gmmiv_set_error("iv_normalize: too many vectors"); return GMMIV_ERR_UNSUPPORTED; }
    if (n == 0) return GMMIV_OK;
    GBIND(c);
    DevIn<double> i_x, i_mu, i_m;
    DevOut<double> o;
    int rc;
    if ((rc = i_x.init(c, WS_T0, X, (size_t)dim_in * n))) return rc;
    if ((rc = i_mu.init(c, WS_T1, mean, dim_in))) return rc;
    if ((rc = i_m.init(c, WS_T2, M, (size_t)dim_out * dim_in))) return rc;
    if ((rc = o.init(c, WS_T3, Y, (size_t)dim_out * n, false))) return rc;
    const double *cur = i_x.d;
    void *p;
    if (mean) { // PldaTest::center (PldaTools.cpp:3754-3767)
        double *dst = o.d;
        if (M || cur == o.d) {
            if ((rc = c->scratch(WS_T4, (size_t)dim_in * n * 8, &p))) return rc;
            dst = (double *)p;
        }
        GCHK(tvk_sub_colvec(c->stream, dim_in, n, cur, i_mu.d, dst));
        cur = dst;
    }
    if (M) { // PldaTest::rotateLeft (:3770-3790): Y = M X
        GCHK(tvk_dgemm(c->stream, false, false, dim_out, (int)n, dim_in, 1.0, i_m.d, dim_in, 0, cur, n, 0, 0.0, o.d, n, 0, 1));
        cur = o.d;
    }
    if (cur != o.d) GCHK(hipMemcpyAsync(o.d, cur, (size_t)dim_out * n * 8, hipMemcpyDeviceToDevice, c->stream));
    if (length_norm) { // PldaTest::lengthNorm (:3706-3751)
        if ((rc = c->scratch(WS_T5, (size_t)n * 8, &p))) return rc;
        GCHK(tvk_coldot(c->stream, dim_out, n, o.d, o.d, (double *)p));
        GCHK(tvk_scale_cols_rsqrt(c->stream, dim_out, n, o.d, (const double *)p));
    }
    return o.finish();
}

// ---- scoring -----------------------------------------------------------------------------
struct ScoreArgs {
    DevIn<double> m, s;
    DevOut<double> sc;
    double *qm = nullptr, *qs = nullptr;
    // row strides of the vector matrices as the GEMMs see them.  _models [dim x M] / _segments [dim x S] have the vector count as
    // their row stride: with an ODD count no row but the first starts on 16 bytes and every GEMM of the rule would run on the
    // per-element checked instantiation (1.5 x slower); such a matrix is copied once into an even-stride block.
    int64_t ldm = 0, lds = 0;
    static int even_stride(gmmiv_ctx *c, int slot, int dim, int64_t n, DevIn<double> &v, int64_t *ld)
    {
        *ld = n;
        if ((n & 1) == 0 || n < 2) return GMMIV_OK;
        void *p;
        int rc = c->scratch(slot, (size_t)dim * (n + 1) * 8, &p);
        if (rc) return rc;
        GCHK(hipMemcpy2DAsync(p, (n + 1) * 8, v.d, n * 8, n * 8, dim, hipMemcpyDeviceToDevice, c->stream));
        v.d = (const double *)p;
        *ld = n + 1;
        return GMMIV_OK;
    }
    int init(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const double *models, const double *segs, double *scores, bool load = false)
    {
        int rc;
        if ((rc = m.init(c, WS_T0, models, (size_t)dim * M))) return rc;
        if ((rc = s.init(c, WS_T1, segs, (size_t)dim * S))) return rc;
        if ((rc = even_stride(c, WS_T9, dim, M, m, &ldm)) || (rc = even_stride(c, WS_TIV, dim, S, s, &lds))) return rc;
        if ((rc = sc.init(c, WS_T2, scores, (size_t)M * S, load))) return rc;
        void *p;
        if ((rc = c->scratch(WS_T3, (size_t)(M + S) * 8, &p))) return rc;
        qm = (double *)p;
        qs = qm + M;
        return GMMIV_OK;
    }
};

static int score_check(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const void *a, const void *b, const void *o, const char *what)
{
    if (!c || dim <= 0 || M < 0 || S < 0 || !a || !b || !o) { gmmiv_set_error("%s: bad argument", what); return GMMIV_ERR_ARG; }
    if (M > 0x7fffffff || S > 0x7fffffff) { gmmiv_set_error("%s: too many vectors", what); return GMMIV_ERR_UNSUPPORTED; }
    GBIND(c);
    return GMMIV_OK;
}

int gmmiv_score_cosine(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const double *models, const double *segs, double *scores)
{
    int rc = score_check(c, dim, M, S, models, segs, scores, "score_cosine");
    if (rc) return rc;
    if (M == 0 || S == 0) return GMMIV_OK;
    ScoreArgs a;
    if ((rc = a.init(c, dim, M, S, models, segs, scores))) return rc;
    GCHK(tvk_coldot(c->stream, dim, M, a.m.d, a.m.d, a.qm, a.ldm));
    GCHK(tvk_coldot(c->stream, dim, S, a.s.d, a.s.d, a.qs, a.lds));
    c->t_begin("k_dgemm(score)");
    GCHK(tvk_rsqrt_vec(c->stream, M, a.qm));   // the normalisation rides in the GEMM epilogue: x 1/|m| x 1/|s|
    GCHK(tvk_rsqrt_vec(c->stream, S, a.qs));
    GCHK(tvk_dgemm_epi(c->stream, true, false, (int)M, (int)S, dim, 1.0, a.m.d, a.ldm, a.s.d, a.lds, a.sc.d, S, 1, a.qm, a.qs, 0.0, 0.0, 0.0));
    c->t_end();
    return a.sc.finish();
}

// scores = Mt (Q + Q^T) S * half_cross + bm * diag(Mt Qm M) + bs * diag(St Qs S)
// ldm: row stride of the model matrix a.m (0: M) -- a run of models gathered by gmmiv_score_plda has an EVEN stride whatever its length
static int quad_score(gmmiv_ctx *c, ScoreArgs &a, int dim, int64_t M, int64_t S, const double *Qcross, double ccross,
                      const double *Qm, double bm, const double *Qs, double bs, double cst, double beta = 0.0, int64_t ldm = 0)
{
    int rc;
    void *p;
    if (ldm <= 0) ldm = a.ldm > 0 ? a.ldm : M;
    const int64_t lds = a.lds > 0 ? a.lds : S;
    const size_t nn = (size_t)dim * dim;
    if ((rc = c->scratch(WS_T4, nn * 8, &p))) return rc;
    double *Qsym = (double *)p;
    const size_t mx = (size_t)dim * (ldm > lds ? ldm : lds);
    if ((rc = c->scratch(WS_T5, mx * 8, &p))) return rc;
    double *Y = (double *)p;
    GCHK(tvk_dgemm(c->stream, false, false, dim, (int)M, dim, 1.0, Qm, dim, 0, a.m.d, ldm, 0, 0.0, Y, ldm, 0, 1));
    GCHK(tvk_coldot(c->stream, dim, M, a.m.d, Y, a.qm, ldm));
    GCHK(tvk_dgemm(c->stream, false, false, dim, (int)S, dim, 1.0, Qs, dim, 0, a.s.d, lds, 0, 0.0, Y, lds, 0, 1));
    GCHK(tvk_coldot(c->stream, dim, S, a.s.d, Y, a.qs, lds));
    GCHK(tvk_add_transpose(c->stream, dim, Qcross, Qcross, Qsym));
    GCHK(tvk_dgemm(c->stream, false, false, dim, (int)S, dim, 1.0, Qsym, dim, 0, a.s.d, lds, 0, 0.0, Y, lds, 0, 1));
    c->t_begin("k_dgemm(score)");
    // ccross m^T Y s + bm q_m + bs q_s + cst in ONE pass over the M x S matrix (GEMM epilogue)
    GCHK(tvk_dgemm_epi(c->stream, true, false, (int)M, (int)S, dim, ccross, a.m.d, ldm, Y, lds, a.sc.d, S, 2, a.qm, a.qs, bm, bs, cst, beta));
    c->t_end();
    return GMMIV_OK;
}

int gmmiv_score_mahalanobis(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const double *models, const double *segs,
                            const double *Mah, double *scores)
{
    int rc = score_check(c, dim, M, S, models, segs, scores, "score_mahalanobis");
    if (rc) return rc;
    if (!Mah) { gmmiv_set_error("score_mahalanobis: Mah == NULL"); return GMMIV_ERR_ARG; }
    if (M == 0 || S == 0) return GMMIV_OK;
    ScoreArgs a;
    if ((rc = a.init(c, dim, M, S, models, segs, scores))) return rc;
    DevIn<double> q;
    if ((rc = q.init(c, WS_T6, Mah, (size_t)dim * dim))) return rc;
    // -1/2 (m-s)' Q (m-s) = -1/2 m'Qm - 1/2 s'Qs + 1/2 m'(Q+Q')s
    if ((rc = quad_score(c, a, dim, M, S, q.d, 0.5, q.d, -0.5, q.d, -0.5, 0.0))) return rc;
    return a.sc.finish();
}

int gmmiv_score_twocov(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const double *models, const double *segs,
                       const double *G, const double *H, double *scores)
{
    int rc = score_check(c, dim, M, S, models, segs, scores, "score_twocov");
    if (rc) return rc;
    if (!G || !H) { gmmiv_set_error("score_twocov: G/H == NULL"); return GMMIV_ERR_ARG; }
    if (M == 0 || S == 0) return GMMIV_OK;
    ScoreArgs a;
    if ((rc = a.init(c, dim, M, S, models, segs, scores))) return rc;
    DevIn<double> g, h;
    const size_t nn = (size_t)dim * dim;
    if ((rc = g.init(c, WS_T6, G, nn))) return rc;
    if ((rc = h.init(c, WS_T7, H, nn))) return rc;
    void *p;
    if ((rc = c->scratch(WS_T8, nn * 8, &p))) return rc;
    double *GmH = (double *)p; // G - H: (m+s)'G(m+s) - m'Hm - s'Hs = m'(G-H)m + s'(G-H)s + m'(G+G')s
    GCHK(tvk_axpby(c->stream, (long)nn, 1.0, g.d, -1.0, h.d, GmH));
    if ((rc = quad_score(c, a, dim, M, S, g.d, 1.0, GmH, 1.0, GmH, 1.0, 0.0))) return rc;
    return a.sc.finish();
}

// PldaTest::twoCovScoringMixPart (PldaTools.cpp:3923-3949): scores[m][s] += (m + s)^T G (m + s) for every pair (ACCUMULATES,
// like the reference's `_scores(m,s) +=`): m'Gm + s'Gs + m'(G + G')s with the model / segment terms in the GEMM epilogue.
int gmmiv_score_twocov_mix_part(gmmiv_ctx *c, int dim, int64_t M, int64_t S, const double *models, const double *segs, const double *G,
                                double *scores)
{
    int rc = score_check(c, dim, M, S, models, segs, scores, "score_twocov_mix_part");
    if (rc) return rc;
    if (!G) { gmmiv_set_error("score_twocov_mix_part: G == NULL"); return GMMIV_ERR_ARG; }
    if (M == 0 || S == 0) return GMMIV_OK;
    ScoreArgs a;
    if ((rc = a.init(c, dim, M, S, models, segs, scores, true))) return rc;
    DevIn<double> g;
    if ((rc = g.init(c, WS_T6, G, (size_t)dim * dim))) return rc;
    if ((rc = quad_score(c, a, dim, M, S, g.d, 1.0, g.d, 1.0, g.d, 1.0, 0.0, 1.0))) return rc;
    return a.sc.finish();
}

// PldaTest::_trials (PldaTools.cpp:3437, 3591-3620): cosineDistance / mahalanobisDistance only score the listed trials
// (:3871, :3889), the others keep the initial value of _scores (0).  The device computes the whole M x S block in one GEMM;
// this entry point then writes `fill` into every cell whose trial flag is 0.  trials: [M x S] bytes, host or device.
int gmmiv_score_apply_trials(gmmiv_ctx *c, int64_t M, int64_t S, const unsigned char *trials, double fill, double *scores)
{
    if (!c || M < 0 || S < 0 || !trials || !scores) { gmmiv_set_error("score_apply_trials: bad argument"); return GMMIV_ERR_ARG; }
    if (M == 0 || S == 0) return GMMIV_OK;
    GBIND(c);
    DevIn<unsigned char> t;
    DevOut<double> o;
    int rc;
    if ((rc = t.init(c, WS_T0, trials, (size_t)M * S)) || (rc = o.init(c, WS_T2, scores, (size_t)M * S, true))) return rc;
    GCHK(tvk_mask_trials(c->stream, (long)(M * S), t.d, fill, o.d));
    return o.finish();
}

// ---- PldaDev: development-set statistics ---------------------------------------------------------
namespace {
struct DevSet { // device views shared by the gmmiv_dev_* entry points
    DevIn<double> x;
    long *off = nullptr;   // [nspk + 1] session offsets
    int *cls = nullptr;    // [n] speaker of each session
    double *ssum = nullptr, *mean = nullptr, *smean = nullptr;
    std::vector<long> hoff;
    int init(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, const char *what)
    {
        if (!c || dim <= 0 || n <= 0 || nspk <= 0 || !X || !sps) { gmmiv_set_error("%s: bad argument", what); return GMMIV_ERR_ARG; }
        if (gmmiv_is_device_ptr(sps)) { gmmiv_set_error("%s: sessions_per_speaker must be a host array", what); return GMMIV_ERR_ARG; }
        GBIND(c);
        hoff.assign(nspk + 1, 0);
        for (int64_t i = 0; i < nspk; ++i) {
            if (sps[i] <= 0) { gmmiv_set_error("%s: speaker %ld has no session", what, (long)i); return GMMIV_ERR_ARG; }
            hoff[i + 1] = hoff[i] + (long)sps[i];
        }
        if (hoff[nspk] != n) { gmmiv_set_error("%s: sessions_per_speaker sums to %ld, n = %ld", what, hoff[nspk], (long)n); return GMMIV_ERR_ARG; }
        std::vector<int> hc(n);
        for (int64_t i = 0; i < nspk; ++i) for (long s = hoff[i]; s < hoff[i + 1]; ++s) hc[s] = (int)i;
        int rc;
        if ((rc = x.init(c, WS_T0, X, (size_t)dim * n))) return rc;
        void *p;
        if ((rc = c->scratch(WS_SEG, (nspk + 1) * sizeof(long) + n * sizeof(int), &p))) return rc;
        off = (long *)p; cls = (int *)(off + nspk + 1);
        GCHK(hipMemcpyAsync(off, hoff.data(), (nspk + 1) * sizeof(long), hipMemcpyHostToDevice, c->stream));
        GCHK(hipMemcpyAsync(cls, hc.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
        GCHK(hipStreamSynchronize(c->stream)); // hc is a stack-lifetime vector
        if ((rc = c->scratch(WS_T1, ((size_t)2 * dim * nspk + dim) * 8, &p))) return rc;
        ssum = (double *)p; smean = ssum + (size_t)dim * nspk; mean = smean + (size_t)dim * nspk;
        GCHK(tvk_dev_means(c->stream, dim, (long)n, x.d, (long)nspk, off, ssum, mean, smean));
        return GMMIV_OK;
    }
};
// out[dim x dim] = alpha * Y Y^T for Y [dim x m] (row-major, ld = m)
int dev_gram(gmmiv_ctx *c, int dim, long m, const double *Y, double alpha, double *out)
{
    const int nz = tvk_splitk_count(dim, dim, (int)m, c->n_cu);
    void *p;
    int rc;
    if ((rc = c->scratch(WS_SLAB, (size_t)nz * dim * dim * 8, &p))) return rc;
    GCHK(tvk_dgemm_splitk(c->stream, false, true, dim, dim, (int)m, alpha, Y, m, Y, m, 0.0, out, dim, nz, (double *)p));
    return GMMIV_OK;
}
} // namespace

int gmmiv_dev_means(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, double *mean, double *spk_means)
{
    DevSet ds;
    int rc = ds.init(c, dim, n, X, nspk, sps, "dev_means");
    if (rc) return rc;
    DevOut<double> o_m, o_s;
    if ((rc = o_m.init(c, WS_T2, mean, dim, false)) || (rc = o_s.init(c, WS_T3, spk_means, (size_t)dim * nspk, false))) return rc;
    if (mean) GCHK(hipMemcpyAsync(o_m.d, ds.mean, dim * 8, hipMemcpyDeviceToDevice, c->stream));
    if (spk_means) GCHK(hipMemcpyAsync(o_s.d, ds.smean, (size_t)dim * nspk * 8, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = o_m.finish())) return rc;
    return o_s.finish();
}

int gmmiv_dev_cov_mat(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, double *Sigma, double *W, double *B)
{
    DevSet ds;
    int rc = ds.init(c, dim, n, X, nspk, sps, "dev_cov_mat");
    if (rc) return rc;
    const size_t dd = (size_t)dim * dim;
    DevOut<double> o_s, o_w, o_b;
    if ((rc = o_s.init(c, WS_T2, Sigma, dd, false)) || (rc = o_w.init(c, WS_T3, W, dd, false)) || (rc = o_b.init(c, WS_T4, B, dd, false))) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)dim * (n > nspk ? n : nspk) * 8, &p))) return rc;
    double *Y = (double *)p;
    const double inv_n = 1.0 / (double)n;
    if (Sigma) {
        GCHK(tvk_dev_center(c->stream, dim, (long)n, 0, ds.x.d, ds.mean, ds.smean, (long)nspk, ds.off, ds.cls, Y));
        if ((rc = dev_gram(c, dim, (long)n, Y, inv_n, o_s.d))) return rc;
    }
    if (W) {
        GCHK(tvk_dev_center(c->stream, dim, (long)n, 1, ds.x.d, ds.mean, ds.smean, (long)nspk, ds.off, ds.cls, Y));
        if ((rc = dev_gram(c, dim, (long)n, Y, inv_n, o_w.d))) return rc;
    }
    if (B) {
        GCHK(tvk_dev_between(c->stream, dim, (long)nspk, 1, ds.mean, ds.smean, ds.off, Y));
        if ((rc = dev_gram(c, dim, (long)nspk, Y, inv_n, o_b.d))) return rc;
    }
    if ((rc = o_s.finish()) || (rc = o_w.finish())) return rc;
    return o_b.finish();
}

int gmmiv_dev_mahalanobis(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, double *M)
{
    if (!M) { gmmiv_set_error("dev_mahalanobis: bad argument"); return GMMIV_ERR_ARG; }
    DevSet ds;
    int rc = ds.init(c, dim, n, X, nspk, sps, "dev_mahalanobis");
    if (rc) return rc;
    DevOut<double> o;
    if ((rc = o.init(c, WS_T2, M, (size_t)dim * dim, false))) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)dim * n * 8, &p))) return rc;
    InvWs ws;
    if ((rc = ws.init(c, dim, 1))) return rc;
    GCHK(tvk_dev_center(c->stream, dim, (long)n, 1, ds.x.d, ds.mean, ds.smean, (long)nspk, ds.off, ds.cls, (double *)p));
    if ((rc = dev_gram(c, dim, (long)n, (double *)p, 1.0 / (double)n, ws.full))) return rc;
    GCHK(hipMemsetAsync(ws.status, 0, sizeof(int), c->stream));
    GCHK(tvk_spd_inverse_batched(c->stream, dim, 1, ws.full, o.d, ws.X, ws.invd, ws.panel, ws.status));
    if ((rc = check_status(c, ws.status, 1, "dev_mahalanobis: W"))) return rc;
    return o.finish();
}

int gmmiv_dev_wccn_chol(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, double *WCCN)
{
    if (!WCCN) { gmmiv_set_error("dev_wccn_chol: bad argument"); return GMMIV_ERR_ARG; }
    DevSet ds;
    int rc = ds.init(c, dim, n, X, nspk, sps, "dev_wccn_chol");
    if (rc) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)dim * n * 8, &p))) return rc;
    InvWs ws;
    if ((rc = ws.init(c, dim, 1))) return rc;
    GCHK(tvk_dev_center(c->stream, dim, (long)n, 2, ds.x.d, ds.mean, ds.smean, (long)nspk, ds.off, ds.cls, (double *)p));
    if ((rc = dev_gram(c, dim, (long)n, (double *)p, 1.0 / (double)nspk, ws.full))) return rc;
    GCHK(hipMemsetAsync(ws.status, 0, sizeof(int), c->stream));
    GCHK(tvk_spd_inverse_batched(c->stream, dim, 1, ws.full, ws.inv, ws.X, ws.invd, ws.panel, ws.status));
    if ((rc = check_status(c, ws.status, 1, "dev_wccn_chol: W"))) return rc;
    std::vector<double> iw, ch; // upperCholesky on the host (O(dim^3) once, like min-divergence)
    if ((rc = fetch_host(c, ws.inv, (size_t)dim * dim, iw))) return rc;
    if (!host_cholesky_upper(dim, iw, ch)) { gmmiv_set_error("dev_wccn_chol: W^-1 is not positive definite"); return GMMIV_ERR_NUMERIC; }
    return store_out(c, WCCN, ch);
}

int gmmiv_dev_scatter_mat(gmmiv_ctx *c, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sps, double *SB, double *SW)
{
    DevSet ds;
    int rc = ds.init(c, dim, n, X, nspk, sps, "dev_scatter_mat");
    if (rc) return rc;
    const size_t dd = (size_t)dim * dim;
    DevOut<double> o_b, o_w;
    if ((rc = o_b.init(c, WS_T2, SB, dd, false)) || (rc = o_w.init(c, WS_T3, SW, dd, false))) return rc;
    void *p;
    if ((rc = c->scratch(WS_TIV, (size_t)dim * (n > nspk ? n : nspk) * 8, &p))) return rc;
    double *Y = (double *)p;
    if (SB) {
        GCHK(tvk_dev_between(c->stream, dim, (long)nspk, 0, ds.mean, ds.smean, ds.off, Y));
        if ((rc = dev_gram(c, dim, (long)nspk, Y, 1.0, o_b.d))) return rc;
    }
    if (SW) { // the reference's loop: the first n_last sessions of the set, centred per speaker, / n_last
        const long nl = (long)sps[nspk - 1];
        GCHK(tvk_dev_center(c->stream, dim, (long)n, 1, ds.x.d, ds.mean, ds.smean, (long)nspk, ds.off, ds.cls, Y));
        const int nz = tvk_splitk_count(dim, dim, (int)nl, c->n_cu);
        if ((rc = c->scratch(WS_SLAB, (size_t)nz * dd * 8, &p))) return rc;
        GCHK(tvk_dgemm_splitk(c->stream, false, true, dim, dim, (int)nl, 1.0 / (double)nl, Y, (long)n, Y, (long)n, 0.0, o_w.d, dim, nz, (double *)p));
    }
    if ((rc = o_b.finish())) return rc;
    return o_w.finish();
}

int gmmiv_sym_eigen(gmmiv_ctx *c, int n, const double *A, int rank, double *vect, double *val)
{
    if (!c || n <= 0 || rank <= 0 || rank > n || !A) { gmmiv_set_error("sym_eigen: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    std::vector<double> a, v, l;
    int rc;
    if ((rc = fetch_host(c, A, (size_t)n * n, a))) return rc;
    host_sym_eigen(n, a, rank, v, l);
    if ((rc = store_out(c, vect, v))) return rc;
    return store_out(c, val, l);
}

int gmmiv_dev_efr_matrix(gmmiv_ctx *c, int dim, const double *Cov, double *M)
{
    if (!c || dim <= 0 || !Cov || !M) { gmmiv_set_error("dev_efr_matrix: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    std::vector<double> a, v, l, m((size_t)dim * dim);
    int rc;
    if ((rc = fetch_host(c, Cov, (size_t)dim * dim, a))) return rc;
    host_sym_eigen(dim, a, dim, v, l);
    for (int j = 0; j < dim; ++j) {
        if (!(l[j] > 0.0)) { gmmiv_set_error("dev_efr_matrix: eigenvalue %d = %g is not positive", j, l[j]); return GMMIV_ERR_NUMERIC; }
        for (int k = 0; k < dim; ++k) m[(size_t)j * dim + k] = v[(size_t)k * dim + j] / sqrt(l[j]);
    }
    return store_out(c, M, m);
}

int gmmiv_dev_lda(gmmiv_ctx *c, int dim, const double *W, const double *B, int rank, double *ldaMat, double *eigval)
{
    if (!c || dim <= 0 || rank <= 0 || rank > dim || !W || !B || !ldaMat) { gmmiv_set_error("dev_lda: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    std::vector<double> w, b, U;
    int rc;
    if ((rc = fetch_host(c, W, (size_t)dim * dim, w)) || (rc = fetch_host(c, B, (size_t)dim * dim, b))) return rc;
    if (!host_cholesky_upper(dim, w, U)) { gmmiv_set_error("dev_lda: W is not positive definite"); return GMMIV_ERR_NUMERIC; }
    // symmetric form of W^-1 B: Cm = L^-1 B L^-T with W = L L^T, L = U^T
    std::vector<double> T1((size_t)dim * dim), Cm((size_t)dim * dim), vect, val, out((size_t)rank * dim);
    for (int j = 0; j < dim; ++j)
        for (int i = 0; i < dim; ++i) {
            double v = b[(size_t)i * dim + j];
            for (int k = 0; k < i; ++k) v -= U[(size_t)k * dim + i] * T1[(size_t)k * dim + j];
            T1[(size_t)i * dim + j] = v / U[(size_t)i * dim + i];
        }
    for (int j = 0; j < dim; ++j)
        for (int i = 0; i < dim; ++i) {
            double v = T1[(size_t)j * dim + i];
            for (int k = 0; k < i; ++k) v -= U[(size_t)k * dim + i] * Cm[(size_t)j * dim + k];
            Cm[(size_t)j * dim + i] = v / U[(size_t)i * dim + i];
        }
    for (int i = 0; i < dim; ++i)
        for (int j = i + 1; j < dim; ++j) { const double m = 0.5 * (Cm[(size_t)i * dim + j] + Cm[(size_t)j * dim + i]); Cm[(size_t)i * dim + j] = Cm[(size_t)j * dim + i] = m; }
    host_sym_eigen(dim, Cm, rank, vect, val);
    for (int j = 0; j < rank; ++j) { // v = U^-1 y, unit norm (Eigen::EigenSolver normalises its eigenvectors)
        double nrm = 0.0;
        for (int i = dim - 1; i >= 0; --i) {
            double v = vect[(size_t)i * rank + j];
            for (int k = i + 1; k < dim; ++k) v -= U[(size_t)i * dim + k] * out[(size_t)j * dim + k];
            out[(size_t)j * dim + i] = v / U[(size_t)i * dim + i];
        }
        for (int i = 0; i < dim; ++i) nrm += out[(size_t)j * dim + i] * out[(size_t)j * dim + i];
        nrm = sqrt(nrm);
        for (int i = 0; i < dim; ++i) out[(size_t)j * dim + i] /= nrm;
    }
    if ((rc = store_out(c, ldaMat, out))) return rc;
    return store_out(c, eigval, val);
}

namespace {
// C[M x N] (+)= op(A) op(B) on the host, op(A) is M x K; row-major, i-k-j order
void hmm(int M, int N, int K, const double *A, bool ta, const double *B, bool tb, double *Cm, bool accumulate = false)
{
    if (!accumulate) memset(Cm, 0, sizeof(double) * (size_t)M * N);
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < K; ++k) {
            const double a = ta ? A[(size_t)k * M + i] : A[(size_t)i * K + k];
            if (a == 0.0) continue;
            double *cr = Cm + (size_t)i * N;
            if (!tb) { const double *br = B + (size_t)k * N; for (int j = 0; j < N; ++j) cr[j] += a * br[j]; }
            else for (int j = 0; j < N; ++j) cr[j] += a * B[(size_t)j * K + k];
        }
}
} // namespace

int gmmiv_plda_em_iteration(gmmiv_ctx *c, int dim, int64_t n, double *X, int64_t nspk, const int64_t *sps, int rf, int rg, double *Fm,
                            double *Gm, double *Sigma, double *Delta)
{
    // rg == 0 (pldaEigenChannelNumber 0, the common "simplified PLDA" configuration): every G-sized object is empty
    if (rf <= 0 || rg < 0 || !Fm || (rg > 0 && !Gm) || !Sigma || !Delta) { gmmiv_set_error("plda_em_iteration: bad argument"); return GMMIV_ERR_ARG; }
    DevSet ds; // validates the arguments, uploads X (when it is a host array), builds cls / off
    int rc = ds.init(c, dim, n, X, nspk, sps, "plda_em_iteration");
    if (rc) return rc;
    const int rh = rf + rg;
    const size_t dd = (size_t)dim * dim;
    hipStream_t st = c->stream;
    std::vector<double> F, G, Sg, Dl;
    if ((rc = fetch_host(c, Fm, (size_t)dim * rf, F)) || (rc = fetch_host(c, Gm, (size_t)dim * rg, G)) || (rc = fetch_host(c, Sigma, dd, Sg)) ||
        (rc = fetch_host(c, Delta, dim, Dl))) return rc;
    // device scratch: centred X (in place when X is a device array), small operands, Eh
    void *p;
    double *Xd = const_cast<double *>(ds.x.d); // DevIn's staging copy or the caller's device array
    if ((rc = c->scratch(WS_T2, ((size_t)rh * dim + (size_t)rg * rg + (size_t)rh * nspk + dim + dd + (size_t)rh * rh + (size_t)dim * rh) * 8, &p))) return rc;
    double *dFG = (double *)p, *dIGG = dFG + (size_t)rh * dim, *dH = dIGG + (size_t)rg * rg, *dDelta = dH + (size_t)rh * nspk;
    double *dOut = dDelta + dim; // sigObs [dd] | gram [rh x rh] | xh [dim x rh]
    if ((rc = c->scratch(WS_TIV, (size_t)2 * rh * n * 8, &p))) return rc;
    double *FGX = (double *)p, *Eh = FGX + (size_t)rh * n; // [rh x n] each: (Ftw; Gtw) X, then the expected latent variables
    // 1. centre by Delta, total second moment
    GCHK(hipMemcpyAsync(dDelta, Dl.data(), dim * 8, hipMemcpyHostToDevice, st));
    GCHK(tvk_sub_colvec(st, dim, (long)n, Xd, dDelta, Xd));
    if ((rc = dev_gram(c, dim, (long)n, Xd, 1.0, dOut))) return rc;
    // 2. preComputation on the host (PldaTools.cpp:2950-2972)
    std::vector<double> Si, FGtw((size_t)rh * dim), GtwG((size_t)rg * rg), iGG, FtwG((size_t)rf * rg), FtwF((size_t)rf * rf), S((size_t)rg * rf),
        A((size_t)rf * rf), t1((size_t)rf * rg);
    if (!host_spd_inverse(dim, Sg, Si, nullptr)) { gmmiv_set_error("plda_em_iteration: Sigma is not positive definite"); return GMMIV_ERR_NUMERIC; }
    double *Ftw = FGtw.data(), *Gtw = FGtw.data() + (size_t)rf * dim;
    hmm(rf, dim, dim, F.data(), true, Si.data(), false, Ftw);
    hmm(rg, dim, dim, G.data(), true, Si.data(), false, Gtw);
    hmm(rg, rg, dim, Gtw, false, G.data(), false, GtwG.data());
    hmm(rf, rg, dim, Ftw, false, G.data(), false, FtwG.data());
    for (int i = 0; i < rg; ++i) GtwG[(size_t)i * rg + i] += 1.0;
    if (!host_spd_inverse(rg, GtwG, iGG, nullptr)) { gmmiv_set_error("plda_em_iteration: G^T S^-1 G + I is not positive definite"); return GMMIV_ERR_NUMERIC; }
    hmm(rf, rf, dim, Ftw, false, F.data(), false, FtwF.data());
    hmm(rg, rf, rg, iGG.data(), false, FtwG.data(), true, S.data());
    hmm(rf, rg, rg, FtwG.data(), false, iGG.data(), false, t1.data());
    hmm(rf, rf, rg, t1.data(), false, FtwG.data(), true, A.data());
    for (size_t i = 0; i < A.size(); ++i) A[i] = FtwF[i] - A[i];
    // 3. (Ftw; Gtw) X on the device, per-speaker sums back to the host
    GCHK(hipMemcpyAsync(dFG, FGtw.data(), FGtw.size() * 8, hipMemcpyHostToDevice, st));
    if (rg > 0) GCHK(hipMemcpyAsync(dIGG, iGG.data(), iGG.size() * 8, hipMemcpyHostToDevice, st));
    GCHK(tvk_dgemm(st, false, false, rh, (int)n, dim, 1.0, dFG, dim, 0, Xd, (long)n, 0, 0.0, FGX, (long)n, 0, 1));
    void *q;
    if ((rc = c->scratch(WS_T3, ((size_t)2 * rh * nspk + rh) * 8, &q))) return rc;
    double *dsum = (double *)q, *dsm = dsum + (size_t)rh * nspk, *dmn = dsm + (size_t)rh * nspk;
    GCHK(tvk_dev_means(st, rh, (long)n, FGX, (long)nspk, ds.off, dsum, dmn, dsm));
    std::vector<double> fg;
    if ((rc = fetch_host(c, dsum, (size_t)rh * nspk, fg))) return rc; // rows 0..rf-1: f_s, rows rf..: g_s
    // 4. per-speaker expectations on the host (:2417-2477)
    std::vector<double> Hs((size_t)rh * nspk), Ehh((size_t)rh * rh, 0.0), U(rh, 0.0), M, MsT((size_t)rf * rg), SMsT((size_t)rg * rg), tmpM((size_t)rh * rh),
        J((size_t)rf * rf), v(rf), gsum(rg, 0.0);
    std::map<int64_t, std::pair<std::vector<double>, std::vector<double> > > cache; // session count -> (M, tmpM)
    for (int64_t spk = 0; spk < nspk; ++spk) {
        const int64_t ns = sps[spk];
        auto it = cache.find(ns);
        if (it == cache.end()) {
            for (size_t i = 0; i < J.size(); ++i) J[i] = (double)ns * A[i];
            for (int i = 0; i < rf; ++i) J[(size_t)i * rf + i] += 1.0;
            if (!host_spd_inverse(rf, J, M, nullptr)) { gmmiv_set_error("plda_em_iteration: n A + I is not positive definite"); return GMMIV_ERR_NUMERIC; }
            hmm(rf, rg, rf, M.data(), false, S.data(), true, MsT.data());
            hmm(rg, rg, rf, S.data(), false, MsT.data(), false, SMsT.data());
            for (int i = 0; i < rf; ++i) for (int j = 0; j < rf; ++j) tmpM[(size_t)i * rh + j] = M[(size_t)i * rf + j];
            for (int i = 0; i < rf; ++i) for (int j = 0; j < rg; ++j) { tmpM[(size_t)i * rh + rf + j] = -MsT[(size_t)i * rg + j]; tmpM[(size_t)(rf + j) * rh + i] = -MsT[(size_t)i * rg + j]; }
            for (int i = 0; i < rg; ++i) for (int j = 0; j < rg; ++j) tmpM[(size_t)(rf + i) * rh + rf + j] = iGG[(size_t)i * rg + j] + SMsT[(size_t)i * rg + j];
            it = cache.emplace(ns, std::make_pair(M, tmpM)).first;
        }
        const std::vector<double> &Mn = it->second.first, &Tn = it->second.second;
        for (int r = 0; r < rf; ++r) { double a = fg[(size_t)r * nspk + spk]; for (int k = 0; k < rg; ++k) a -= S[(size_t)k * rf + r] * fg[(size_t)(rf + k) * nspk + spk]; v[r] = a; }
        for (int r = 0; r < rf; ++r) { double a = 0.0; for (int k = 0; k < rf; ++k) a += Mn[(size_t)r * rf + k] * v[k]; Hs[(size_t)r * nspk + spk] = a; U[r] += (double)ns * a; }
        for (int r = 0; r < rg; ++r) { double a = 0.0; for (int k = 0; k < rf; ++k) a += S[(size_t)r * rf + k] * Hs[(size_t)k * nspk + spk]; Hs[(size_t)(rf + r) * nspk + spk] = a; U[rf + r] -= (double)ns * a; gsum[r] += fg[(size_t)(rf + r) * nspk + spk]; }
        for (size_t i = 0; i < Ehh.size(); ++i) Ehh[i] += (double)ns * Tn[i];
    }
    for (int r = 0; r < rg; ++r) { double a = 0.0; for (int k = 0; k < rg; ++k) a += iGG[(size_t)r * rg + k] * gsum[k]; U[rf + r] += a; }
    // 5. Eh = [h_spk ; iGG g_i - S h_spk] per session, its Gram matrix and X Eh^T on the device
    GCHK(hipMemcpyAsync(dH, Hs.data(), Hs.size() * 8, hipMemcpyHostToDevice, st));
    GCHK(tvk_dev_expand(st, rf, (long)n, (long)nspk, dH, ds.cls, Eh));
    if (rg > 0) {
        GCHK(tvk_dgemm(st, false, false, rg, (int)n, rg, 1.0, dIGG, rg, 0, FGX + (size_t)rf * n, (long)n, 0, 0.0, Eh + (size_t)rf * n, (long)n, 0, 1));
        GCHK(tvk_dev_center(st, rg, (long)n, 1, Eh + (size_t)rf * n, nullptr, dH + (size_t)rf * nspk, (long)nspk, ds.off, ds.cls, Eh + (size_t)rf * n));
    }
    double *dGram = dOut + dd, *dXh = dGram + (size_t)rh * rh;
    if ((rc = dev_gram(c, rh, (long)n, Eh, 1.0, dGram))) return rc;
    {
        const int nz = tvk_splitk_count(dim, rh, (int)n, c->n_cu);
        if ((rc = c->scratch(WS_SLAB, (size_t)nz * dim * rh * 8, &q))) return rc;
        GCHK(tvk_dgemm_splitk(st, false, true, dim, rh, (int)n, 1.0, Xd, (long)n, Eh, (long)n, 0.0, dXh, rh, nz, (double *)q));
    }
    std::vector<double> outv;
    if ((rc = fetch_host(c, dOut, dd + (size_t)rh * rh + (size_t)dim * rh, outv))) return rc;
    const double *sigObs = outv.data(), *gram = sigObs + dd, *xh = gram + (size_t)rh * rh;
    for (size_t i = 0; i < Ehh.size(); ++i) Ehh[i] += gram[i];
    // 6. mStep on the host (:2790-2815)
    std::vector<double> iE, FG((size_t)dim * rh), SL(dd), cF((size_t)rf * rf), cG((size_t)rg * rg), Rh, Rw;
    if (!host_spd_inverse(rh, Ehh, iE, nullptr)) { gmmiv_set_error("plda_em_iteration: sum E[hh^T] is not positive definite"); return GMMIV_ERR_NUMERIC; }
    hmm(dim, rh, rh, xh, false, iE.data(), false, FG.data());
    hmm(dim, dim, rh, FG.data(), false, xh, true, SL.data());
    for (size_t i = 0; i < dd; ++i) Sg[i] = (sigObs[i] - SL[i]) / (double)n;
    for (int i = 0; i < rh; ++i) U[i] /= (double)n;
    for (int i = 0; i < rf; ++i) for (int j = 0; j < rf; ++j) cF[(size_t)i * rf + j] = Ehh[(size_t)i * rh + j] / (double)n - U[i] * U[j];
    for (int i = 0; i < rg; ++i) for (int j = 0; j < rg; ++j) cG[(size_t)i * rg + j] = Ehh[(size_t)(rf + i) * rh + rf + j] / (double)n - U[rf + i] * U[rf + j];
    if (!host_cholesky_upper(rf, cF, Rh) || !host_cholesky_upper(rg, cG, Rw)) { gmmiv_set_error("plda_em_iteration: minimum-divergence covariance is not positive definite"); return GMMIV_ERR_NUMERIC; }
    for (int i = 0; i < dim; ++i) {
        for (int j = 0; j < rf; ++j) { double a = 0.0; for (int k = 0; k < rf; ++k) a += FG[(size_t)i * rh + k] * Rh[(size_t)j * rf + k]; F[(size_t)i * rf + j] = a; }
        for (int j = 0; j < rg; ++j) { double a = 0.0; for (int k = 0; k < rg; ++k) a += FG[(size_t)i * rh + rf + k] * Rw[(size_t)j * rg + k]; G[(size_t)i * rg + j] = a; }
        double d = 0.0;
        for (int k = 0; k < rh; ++k) d += FG[(size_t)i * rh + k] * U[k];
        Dl[i] += d;
    }
    if ((rc = store_out(c, Fm, F)) || (rc = store_out(c, Gm, G)) || (rc = store_out(c, Sigma, Sg)) || (rc = store_out(c, Delta, Dl))) return rc;
    if (!gmmiv_is_device_ptr(X)) { // the centred data goes back to the caller's host array
        GCHK(hipMemcpyAsync(X, Xd, (size_t)dim * n * 8, hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
    }
    return GMMIV_OK;
}

int gmmiv_plda_precompute(gmmiv_ctx *c, int dim, int rf, int rg, const double *Fm, const double *Gm, const double *Sigma, double *FTJ,
                          double *FTJF)
{
    if (!c || dim <= 0 || rf <= 0 || rg < 0 || !Fm || (rg > 0 && !Gm) || !Sigma || !FTJ || !FTJF) { gmmiv_set_error("plda_precompute: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    DevIn<double> i_f, i_g, i_s;
    DevOut<double> o_j, o_jf;
    int rc;
    if ((rc = i_f.init(c, WS_T0, Fm, (size_t)dim * rf)) || (rc = i_g.init(c, WS_T1, Gm, (size_t)dim * rg)) || (rc = i_s.init(c, WS_T2, Sigma, (size_t)dim * dim)) ||
        (rc = o_j.init(c, WS_T3, FTJ, (size_t)rf * dim, false)) || (rc = o_jf.init(c, WS_T9, FTJF, (size_t)rf * rf, false))) return rc;
    const int big = dim > rg ? dim : rg;
    InvWs ws;
    if ((rc = ws.init(c, big, 1))) return rc;
    void *p;
    const size_t need = (size_t)dim * dim + (size_t)rf * dim + (size_t)rg * dim + (size_t)rg * rg * 2 + (size_t)rf * rg * 2;
    if ((rc = c->scratch(WS_AUX, need * 8, &p))) return rc;
    double *Si = (double *)p, *Ftw = Si + (size_t)dim * dim, *Gtw = Ftw + (size_t)rf * dim, *GG = Gtw + (size_t)rg * dim;
    double *Mi = GG + (size_t)rg * rg, *FtwG = Mi + (size_t)rg * rg, *t1 = FtwG + (size_t)rf * rg;
    hipStream_t st = c->stream;
    // S^-1 (the inverse routine factors its input in place: work on a copy)
    GCHK(hipMemcpyAsync(ws.full, i_s.d, (size_t)dim * dim * 8, hipMemcpyDeviceToDevice, st));
    GCHK(hipMemsetAsync(ws.status, 0, sizeof(int), st));
    GCHK(tvk_spd_inverse_batched(st, dim, 1, ws.full, Si, ws.X, ws.invd, ws.panel, ws.status));
    if ((rc = check_status(c, ws.status, 1, "plda_precompute: Sigma"))) return rc;
    GCHK(tvk_dgemm(st, true, false, rf, dim, dim, 1.0, i_f.d, rf, 0, Si, dim, 0, 0.0, Ftw, dim, 0, 1));          // F^T S^-1
    GCHK(hipMemcpyAsync(o_j.d, Ftw, (size_t)rf * dim * 8, hipMemcpyDeviceToDevice, st));
    if (rg > 0) {
        GCHK(tvk_dgemm(st, true, false, rg, dim, dim, 1.0, i_g.d, rg, 0, Si, dim, 0, 0.0, Gtw, dim, 0, 1));      // G^T S^-1
        GCHK(tvk_dgemm(st, false, false, rg, rg, dim, 1.0, Gtw, dim, 0, i_g.d, rg, 0, 0.0, GG, rg, 0, 1));        // G^T S^-1 G
        GCHK(tvk_add_identity(st, rg, GG));
        GCHK(tvk_dgemm(st, false, false, rf, rg, dim, 1.0, Ftw, dim, 0, i_g.d, rg, 0, 0.0, FtwG, rg, 0, 1));      // F^T S^-1 G
        GCHK(hipMemcpyAsync(ws.full, GG, (size_t)rg * rg * 8, hipMemcpyDeviceToDevice, st));
        GCHK(hipMemsetAsync(ws.status, 0, sizeof(int), st));
        GCHK(tvk_spd_inverse_batched(st, rg, 1, ws.full, Mi, ws.X, ws.invd, ws.panel, ws.status));
        if ((rc = check_status(c, ws.status, 1, "plda_precompute: G^T S^-1 G + I"))) return rc;
        GCHK(tvk_dgemm(st, false, false, rf, rg, rg, 1.0, FtwG, rg, 0, Mi, rg, 0, 0.0, t1, rg, 0, 1));
        GCHK(tvk_dgemm(st, false, false, rf, dim, rg, -1.0, t1, rg, 0, Gtw, dim, 0, 1.0, o_j.d, dim, 0, 1));       // FTJ -= t1 Gtw
    }
    GCHK(tvk_dgemm(st, false, false, rf, rf, dim, 1.0, o_j.d, dim, 0, i_f.d, rf, 0, 0.0, o_jf.d, rf, 0, 1));
    if ((rc = o_j.finish())) return rc;
    return o_jf.finish();
}

int gmmiv_twocov_model(gmmiv_ctx *c, int dim, const double *W, const double *B, double *G, double *H)
{
    if (!c || dim <= 0 || !W || !B || !G || !H) { gmmiv_set_error("twocov_model: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t dd = (size_t)dim * dim;
    DevIn<double> i_w, i_b;
    DevOut<double> o_g, o_h;
    int rc;
    if ((rc = i_w.init(c, WS_T0, W, dd)) || (rc = i_b.init(c, WS_T1, B, dd)) || (rc = o_g.init(c, WS_T2, G, dd, false)) || (rc = o_h.init(c, WS_T3, H, dd, false))) return rc;
    InvWs ws;
    if ((rc = ws.init(c, dim, 1))) return rc;
    void *p;
    if ((rc = c->scratch(WS_AUX, 5 * dd * 8, &p))) return rc;
    double *iW = (double *)p, *iB = iW + dd, *sm = iB + dd, *ti = sm + dd, *t2 = ti + dd;
    hipStream_t st = c->stream;
    auto inv = [&](const double *src, double *dst, const char *what) -> int {
        GCHK(hipMemcpyAsync(ws.full, src, dd * 8, hipMemcpyDeviceToDevice, st));
        GCHK(hipMemsetAsync(ws.status, 0, sizeof(int), st));
        GCHK(tvk_spd_inverse_batched(st, dim, 1, ws.full, dst, ws.X, ws.invd, ws.panel, ws.status));
        return check_status(c, ws.status, 1, what);
    };
    if ((rc = inv(i_w.d, iW, "twocov_model: W")) || (rc = inv(i_b.d, iB, "twocov_model: B"))) return rc;
    for (int pass = 0; pass < 2; ++pass) { // G: B^-1 + 2 W^-1 ; H: B^-1 + W^-1
        GCHK(tvk_axpby(st, (long)dd, 1.0, iB, pass == 0 ? 2.0 : 1.0, iW, sm));
        if ((rc = inv(sm, ti, "twocov_model: B^-1 + a W^-1"))) return rc;
        GCHK(tvk_dgemm(st, false, false, dim, dim, dim, 1.0, iW, dim, 0, ti, dim, 0, 0.0, t2, dim, 0, 1));
        GCHK(tvk_dgemm(st, false, false, dim, dim, dim, 1.0, t2, dim, 0, iW, dim, 0, 0.0, pass == 0 ? o_g.d : o_h.d, dim, 0, 1));
    }
    if ((rc = o_g.finish())) return rc;
    return o_h.finish();
}

int gmmiv_score_plda(gmmiv_ctx *c, int rf, int64_t M, int64_t S, const double *models_sum, const int64_t *nsess,
                     const double *segs, const double *FTJF, double *scores)
{
    int rc = score_check(c, rf, M, S, models_sum, segs, scores, "score_plda");
    if (rc) return rc;
    if (!nsess || !FTJF) { gmmiv_set_error("score_plda: nsess/FTJF == NULL"); return GMMIV_ERR_ARG; }
    if (gmmiv_is_device_ptr(nsess)) { gmmiv_set_error("score_plda: nsess must be a host array"); return GMMIV_ERR_ARG; }
    if (M == 0 || S == 0) return GMMIV_OK;
    ScoreArgs a;
    if ((rc = a.init(c, rf, M, S, models_sum, segs, scores))) return rc;
    const size_t nn = (size_t)rf * rf;
    std::vector<double> hF(nn);
    GCHK(hipMemcpy(hF.data(), FTJF, nn * 8, gmmiv_is_device_ptr(FTJF) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    // K_n = (n FTJF + I)^-1 and alpha_n = log det K_n on the host, cached per n in the context for as long as FTJF is unchanged
    typedef gmmiv_ctx::PldaK KN;
    if (c->plda_ftjf.size() != nn || memcmp(c->plda_ftjf.data(), hF.data(), nn * 8) != 0) {
        c->plda_ftjf = hF;
        c->plda_k.clear();
    }
    std::map<long, KN> &cache = c->plda_k;
    auto getK = [&](int64_t n) -> const KN * {
        auto it = cache.find((long)n);
        if (it != cache.end()) return &it->second;
        std::vector<double> t(nn);
        for (size_t e = 0; e < nn; ++e) t[e] = (double)n * hF[e];
        for (int i = 0; i < rf; ++i) t[(size_t)i * rf + i] += 1.0;
        KN kn;
        double ld;
        if (!host_spd_inverse(rf, t, kn.K, &ld)) return nullptr;
        kn.alpha = -ld; // log det K = -log det (nFTJF + I)
        return &cache.emplace((long)n, std::move(kn)).first->second;
    };
    const KN *K1 = getK(1);
    if (!K1) { gmmiv_set_error("score_plda: FTJF + I is not positive definite"); return GMMIV_ERR_NUMERIC; }
    void *p;
    if ((rc = c->scratch(WS_T6, 3 * nn * 8, &p))) return rc;
    double *dQc = (double *)p, *dQm = dQc + nn, *dQs = dQm + nn;
    // runs of consecutive models with the same session count (PldaTools.cpp:4186-4250)
    for (int64_t m0 = 0; m0 < M;) {
        int64_t m1 = m0;
        const int64_t L = nsess[m0];
        while (m1 < M && nsess[m1] == L) ++m1;
        if (L < 1) { gmmiv_set_error("score_plda: nsess[%ld] < 1", (long)m0); return GMMIV_ERR_ARG; }
        const KN *KL = getK(L), *KL1 = getK(L + 1);
        if (!KL || !KL1) { gmmiv_set_error("score_plda: K_n not positive definite"); return GMMIV_ERR_NUMERIC; }
        // score = 1/2[(s+m)'K_{L+1}(s+m) - m'K_L m - s'K_1 s] + (a_{L+1} - a_L - a_1)/2
        //       = 1/2 m'(K_{L+1}-K_L)m + 1/2 s'(K_{L+1}-K_1)s + 1/2 m'(K_{L+1}+K_{L+1}')s + cst
        std::vector<double> qm(nn), qs(nn);
        for (size_t e = 0; e < nn; ++e) { qm[e] = KL1->K[e] - KL->K[e]; qs[e] = KL1->K[e] - K1->K[e]; }
        GCHK(hipMemcpyAsync(dQc, KL1->K.data(), nn * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipMemcpyAsync(dQm, qm.data(), nn * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipMemcpyAsync(dQs, qs.data(), nn * 8, hipMemcpyHostToDevice, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
        const double cst = (KL1->alpha - KL->alpha - K1->alpha) / 2.0;
        // operate on the column range [m0, m1) of models (ld = M) and the row range of scores
        ScoreArgs sub = a;
        const int64_t Mr = m1 - m0;
        // gather the run's columns into a compact block [rf x Mr] with an EVEN row stride: with an odd one (a run of odd length, half of
        // all runs) no row but the first starts on 16 bytes and the whole scoring GEMM fell to the per-element checked instantiation
        // (37 instead of 24 ms per third of 100 k x 100 k trials: 98 G trials/s where 137 are possible)
        void *q;
        const int64_t ldq = Mr + (Mr & 1);
        if ((rc = c->scratch(WS_T7, (size_t)rf * ldq * 8, &q))) return rc;
        GCHK(hipMemcpy2DAsync(q, ldq * 8, a.m.d + m0, a.ldm * 8, Mr * 8, rf, hipMemcpyDeviceToDevice, c->stream));
        sub.m.d = (const double *)q;
        sub.sc.d = a.sc.d + (size_t)m0 * S;
        sub.qm = a.qm + m0;
        if ((rc = quad_score(c, sub, rf, Mr, S, dQc, 0.5, dQm, 0.5, dQs, 0.5, cst, 0.0, ldq))) return rc;
        GCHK(hipStreamSynchronize(c->stream));
        m0 = m1;
    }
    return a.sc.finish();
}

} // extern "C"
