// capi_gmm.hip -- C ABI (include/gmmiv.h): context, model, GMM likelihood / statistics entry points.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include "ctx.h"
#include "gmm_kernels.h"
#include "tv_kernels.h"

static thread_local char g_err[512] = "";

// The bound set is a COPY (52 bytes per call): a thread that last drove a context which another thread has destroyed since holds no
// pointer into freed memory; g_kopts_src is kept for the identity test of "kopts_bound" only and never dereferenced.
static thread_local gmmiv_kopts g_kopts_val;
static thread_local const gmmiv_kopts *g_kopts_src = nullptr;
const gmmiv_kopts &gmmiv_kopts_cur() { return g_kopts_val; }
void gmmiv_kopts_bind(const gmmiv_kopts *ko) { g_kopts_val = ko ? *ko : gmmiv_kopts(); g_kopts_src = ko; }

void gmmiv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool gmmiv_is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // plain malloc'ed host memory on older runtimes
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

extern "C" {

const char *gmmiv_last_error(void) { return g_err; }
const char *gmmiv_version(void) { return "gmmiv 0.1 (gfx950)"; }

int gmmiv_ctx_create(int device, void *stream, gmmiv_ctx **out)
{
    if (!out) { gmmiv_set_error("ctx_create: out == NULL"); return GMMIV_ERR_ARG; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        gmmiv_set_error("ctx_create: no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return GMMIV_ERR_HIP;
    }
    if (device < 0 || device >= ndev) { gmmiv_set_error("ctx_create: device %d out of range [0,%d)", device, ndev); return GMMIV_ERR_ARG; }
    GCHK(hipSetDevice(device));
    gmmiv_ctx *c = new gmmiv_ctx();
    c->device = device;
    if (stream == GMMIV_STREAM_DEFAULT || stream == (void *)hipStreamLegacy) { c->stream = nullptr; c->own_stream = false; } // the NULL stream itself
    else if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { GCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) { c->n_cu = prop.multiProcessorCount; c->total_mem = prop.totalGlobalMem; }
    if (hipMalloc((void **)&c->d_zero_llk, 2 * sizeof(unsigned long long)) != hipSuccess || hipMemset(c->d_zero_llk, 0, 2 * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        gmmiv_set_error("ctx_create: hipMalloc of the context's device counters failed");
        return GMMIV_ERR_HIP;
    }
    c->d_screened = c->d_zero_llk + 1;
    *out = c;
    return GMMIV_OK;
}

void gmmiv_ctx_destroy(gmmiv_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (gmmiv_comm *cm : std::vector<gmmiv_comm *>(c->comms)) gmmiv_comm_orphan(cm); // their handles stay valid for gmmiv_comm_destroy
    for (int i = 0; i < WS_COUNT; ++i)
        if (c->ws[i]) (void)hipFree(c->ws[i]);
    for (int i = 0; i < gmmiv_ctx::NSLOT; ++i) {
        for (hipEvent_t e : c->ev0[i]) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->ev1[i]) (void)hipEventDestroy(e);
    }
    if (c->d_zero_llk) (void)hipFree(c->d_zero_llk);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    if (g_kopts_src == &c->ko) gmmiv_kopts_bind(nullptr); // back to the defaults on this thread
    delete c;
}

int gmmiv_ctx_sync(gmmiv_ctx *c)
{
    if (!c) return GMMIV_ERR_ARG;
    GCHK(hipStreamSynchronize(c->stream));
    return GMMIV_OK;
}

void *gmmiv_ctx_stream(gmmiv_ctx *c) { return c ? (void *)c->stream : nullptr; }

long gmmiv_ctx_set_option(gmmiv_ctx *c, const char *key, long value)
{
    if (!c || !key) return -1;
    long *slot = nullptr;
    if (!strcmp(key, "glds")) slot = &c->use_glds;
    else if (!strcmp(key, "em_chunks")) slot = &c->em_chunks;
    else if (!strcmp(key, "timing")) slot = &c->timing;
    else if (!strcmp(key, "wg_waves")) slot = &c->wg_waves;
    else if (!strcmp(key, "dbg")) slot = &c->dbg;
    else if (!strcmp(key, "prune_log2")) slot = &c->prune_log2;
    else if (!strcmp(key, "stats_z")) slot = &c->stats_z;
    else if (!strcmp(key, "z_scratch_mb")) slot = &c->z_scratch_mb;
    else if (!strcmp(key, "tv_batch")) slot = &c->tv_batch;
    else if (!strcmp(key, "tv_tett_direct")) slot = &c->tv_tett_direct;
    else if (!strcmp(key, "tv_stats_split")) slot = &c->tv_stats_split;
    else if (!strcmp(key, "topc_z")) slot = &c->topc_z;
    else if (!strcmp(key, "tv_mstep_solve")) slot = &c->tv_mstep_solve;
    else if (!strcmp(key, "tv_md_device")) slot = &c->tv_md_device;
    else if (!strcmp(key, "tv_acc_mb")) slot = &c->tv_acc_mb;
    else if (!strcmp(key, "topc_fused")) slot = &c->topc_fused;
    else if (!strcmp(key, "topc_fallbacks")) slot = &c->topc_fallbacks;
    else if (!strcmp(key, "topc_rank_direct")) slot = &c->topc_rank_direct;
    else if (!strcmp(key, "topc_rank2")) slot = &c->topc_rank2;
    else if (!strcmp(key, "topc_use_lanes")) slot = &c->topc_use_lanes;
    else if (!strcmp(key, "assume_finite")) slot = &c->assume_finite;
    // options read by the kernel launchers: kept in the context's gmmiv_kopts, bound to the calling thread by every call (GBIND)
    int *ks = nullptr;
    if (!strcmp(key, "z_waves")) { const long prev = c->ko.z_waves; c->ko.z_waves = (value == 4 || value == 16) ? (int)value : 8; return prev; }
    if (!strcmp(key, "z_depth_em")) { const long prev = c->ko.z_depth_em; if (value == 2 || value == 4) c->ko.z_depth_em = (int)value; return prev; }
    if (!strcmp(key, "z_depth_tv")) { const long prev = c->ko.z_depth_tv; if (value == 2 || value == 4) c->ko.z_depth_tv = (int)value; return prev; }
    if (!strcmp(key, "z_tv4")) ks = &c->ko.z_tv4;                 // A/B knob: 0 = two Gaussian tiles per wave in the N / F mode of k_stats_z
    else if (!strcmp(key, "gemm_remap")) ks = &c->ko.gemm_remap;   // A/B knob: 2 = 8 x 8 tile blocks per XCD, 1 = M tiles fastest per XCD (rounds 2-4), 0 = hardware tile order in k_dgemm
    else if (!strcmp(key, "gemm_clamp")) ks = &c->ko.gemm_clamp;   // A/B knob: 0 = cut GEMM tiles on the per-element checked instantiation
    else if (!strcmp(key, "gemm_narrow")) ks = &c->ko.gemm_narrow; // A/B knob: 0 = 128 x 128 tiles on the strips cut by M / N too
    else if (!strcmp(key, "gemm_nt80")) ks = &c->ko.gemm_nt80;     // A/B knob: 0 = no 128 x 80 tiles for N = 5 x 80 (aux at rank 400)
    else if (!strcmp(key, "chol_lds")) ks = &c->ko.chol_lds;       // 0 = panel rows from memory per wave
    else if (!strcmp(key, "chol_gemm")) ks = &c->ko.chol_gemm;     // the GEMM-built batched Cholesky instead of k_chol_left
    else if (!strcmp(key, "chol_waves")) ks = &c->ko.chol_waves;   // 16 = k_trinv_left / k_uut on 1024-thread workgroups
    else if (!strcmp(key, "short_calls")) ks = &c->ko.short_calls; // 0 = calls of at most 32768 frames on the kernel shapes of long calls
    else if (!strcmp(key, "chol_flow")) ks = &c->ko.chol_flow;     // 0 = the round-2 k_chol_left (diagonal update from L2, eight partial blocks)
    if (ks) { const long prev = *ks; *ks = (int)value; return prev; }
    // the device counters of kind-(2) frames (zero likelihood under the call's model -- this INCLUDES the kind-(1) frames, which every
    // kernel evaluates as such) and of kind-(1) frames (unusable feature values, counted by the screening pass): reading one waits
    // for the stream, the counting itself never does
    if (!strcmp(key, "zero_llk_frames") || !strcmp(key, "screened_frames")) {
        unsigned long long *ctr = key[0] == 'z' ? c->d_zero_llk : c->d_screened;
        unsigned long long prev = 0, nv = value > 0 ? (unsigned long long)value : 0;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(&prev, ctr, sizeof prev, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ctr, &nv, sizeof nv, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return -1; }
        return (long)prev;
    }
    if (!strcmp(key, "kopts_bound")) return g_kopts_src == &c->ko ? 1 : 0; // read-only: is this context's set the one bound to the calling thread?
    if (!slot) return -1;
    long prev = *slot;
    *slot = value;
    return prev;
}

int gmmiv_ctx_set_hook(gmmiv_ctx *c, const char *point, gmmiv_hook_fn fn, void *user)
{
    if (!c || !point) return -1;
    gmmiv_ctx::Hook *h = !strcmp(point, "tv_a_ready") ? &c->hook_tv_a_ready : (!strcmp(point, "md_factored") ? &c->hook_md_factored : nullptr);
    if (!h) return -1;
    h->fn = fn;
    h->user = fn ? user : nullptr;
    return 0;
}

double gmmiv_ctx_last_kernel_ms(gmmiv_ctx *c, const char **name)
{
    if (!c || c->ev_last < 0) return -1.0;
    if (name) *name = c->ev_name[c->ev_last];
    return c->t_query(c->ev_last);
}

double gmmiv_ctx_kernel_ms(gmmiv_ctx *c, const char *name)
{
    if (!c || !name) return -1.0;
    for (int i = 0; i < gmmiv_ctx::NSLOT; ++i)
        if (c->ev_name[i] && !strcmp(c->ev_name[i], name)) return c->t_query(i);
    return -1.0;
}

long gmmiv_ctx_kernel_launches(gmmiv_ctx *c, const char *name)
{
    if (!c || !name) return -1;
    for (int i = 0; i < gmmiv_ctx::NSLOT; ++i)
        if (c->ev_name[i] && !strcmp(c->ev_name[i], name)) return c->ev_used[i];
    return -1;
}

// ---- model -------------------------------------------------------------------------------
static int gmm_upload(gmmiv_gmm *g, const double *w, const double *mean, const double *covinv)
{
    gmmiv_ctx *c = g->ctx;
    GBIND(c);
    const size_t CD = (size_t)g->C * g->D;
    auto kind = [](const void *p) { return gmmiv_is_device_ptr(p) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice; };
    GCHK(hipMemcpyAsync(g->w, w, g->C * sizeof(double), kind(w), c->stream));
    GCHK(hipMemcpyAsync(g->mean, mean, CD * sizeof(double), kind(mean), c->stream));
    GCHK(hipMemcpyAsync(g->iv, covinv, CD * sizeof(double), kind(covinv), c->stream));
    GCHK(gmmk_pack_model(c->stream, g->C, g->D, g->KS, g->nct, g->Cp64, g->w, g->mean, g->iv, g->a, g->lwc, g->Pt,
                         g->meanT, g->ivT));
    GCHK(hipStreamSynchronize(c->stream)); // host sources may be freed by the caller on return
    return GMMIV_OK;
}

int gmmiv_gmm_create(gmmiv_ctx *c, int C, int D, const double *w, const double *mean, const double *covinv,
                     gmmiv_gmm **out)
{
    if (!c || !out || !w || !mean || !covinv || C <= 0 || D <= 0) { gmmiv_set_error("gmm_create: bad argument"); return GMMIV_ERR_ARG; }
    const int KS = gmmk_ks_for_dim(D);
    if (!KS) { gmmiv_set_error("gmm_create: vectSize %d not supported (max %d)", D, GMMK_MAX_DIM); return GMMIV_ERR_UNSUPPORTED; }
    // KS == GMMK_KS_GENERIC (vectSize > 80): no packed MFMA model; every entry point takes its generic path (VALU logits in the
    // reference's direct form, statistics as gamma^T [x | 1 | x^2] on the fp64 GEMM) -- slower, same results, nothing refused
    GBIND(c);
    gmmiv_gmm *g = new gmmiv_gmm();
    g->ctx = c; g->C = C; g->D = D; g->KS = KS;
    g->nct = ((C + 15) / 16 + 1) / 2 * 2; // c-tiles of 16, padded to the LLK kernel's stage of 2
    g->Cp64 = (C + 63) / 64 * 64;
    const size_t CD = (size_t)C * D;
    const int Cpa = g->Cp64 > g->nct * 16 ? g->Cp64 : g->nct * 16;
    struct { void **p; size_t n; } need[] = {{(void **)&g->w, (size_t)C}, {(void **)&g->mean, CD}, {(void **)&g->iv, CD}, {(void **)&g->a, (size_t)Cpa},
                                           {(void **)&g->lwc, (size_t)Cpa}, {(void **)&g->Pt, KS == GMMK_KS_GENERIC ? (size_t)64 : (size_t)g->nct * (2 * KS + 2) * 64},
                                           {(void **)&g->meanT, (size_t)D * g->Cp64}, {(void **)&g->ivT, (size_t)D * g->Cp64}};
    for (auto &a : need) {
        const hipError_t e = hipMalloc(a.p, a.n * sizeof(double));
        if (e != hipSuccess) { // nothing allocated so far may leak
            gmmiv_set_error("gmm_create: hipMalloc of %zu bytes -> %s", a.n * sizeof(double), hipGetErrorString(e));
            gmmiv_gmm_destroy(g);
            return GMMIV_ERR_HIP;
        }
    }
    int rc = gmm_upload(g, w, mean, covinv);
    if (rc) { gmmiv_gmm_destroy(g); return rc; }
    *out = g;
    return GMMIV_OK;
}

int gmmiv_gmm_set(gmmiv_gmm *g, const double *w, const double *mean, const double *covinv)
{
    if (!g || !w || !mean || !covinv) { gmmiv_set_error("gmm_set: bad argument"); return GMMIV_ERR_ARG; }
    return gmm_upload(g, w, mean, covinv);
}

int gmmiv_gmm_set_cov(gmmiv_gmm *g, const double *w, const double *mean, const double *cov)
{
    if (!g || !w || !mean || !cov) { gmmiv_set_error("gmm_set_cov: bad argument"); return GMMIV_ERR_ARG; }
    gmmiv_ctx *c = g->ctx;
    GBIND(c);
    const size_t CD = (size_t)g->C * g->D;
    DevIn<double> i_cov;
    int rc = i_cov.init(c, WS_T0, cov, CD);
    if (rc) return rc;
    void *p;
    if ((rc = c->scratch(WS_T1, CD * sizeof(double), &p))) return rc;
    GCHK(gmmk_reciprocal(c->stream, (long)CD, i_cov.d, (double *)p));   // covInv = 1 / cov (DistribGD::computeAll)
    return gmm_upload(g, w, mean, (const double *)p);
}

void gmmiv_gmm_destroy(gmmiv_gmm *g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    void *ptrs[] = {g->w, g->mean, g->iv, g->a, g->lwc, g->Pt, g->meanT, g->ivT};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
}

// ---- helpers -----------------------------------------------------------------------------
static size_t esize(int dt) { return dt == GMMIV_F64 ? 8 : 4; }

// Device view of the feature block [T x ldx]; host input is copied (compacted to ldx = D).
struct XView {
    const void *d = nullptr;
    int64_t ldx = 0;
    int init(gmmiv_ctx *c, const void *x, int dt, int64_t T, int64_t ld, int D)
    {
        if (dt != GMMIV_F32 && dt != GMMIV_F64) { gmmiv_set_error("feature dtype must be GMMIV_F32 or GMMIV_F64"); return GMMIV_ERR_ARG; }
        if (ld < D) { gmmiv_set_error("ldx (%ld) < D (%d)", (long)ld, D); return GMMIV_ERR_ARG; }
        if (T == 0) { d = x; ldx = ld; return GMMIV_OK; }
        if (!x) { gmmiv_set_error("x == NULL"); return GMMIV_ERR_ARG; }
        if (gmmiv_is_device_ptr(x)) { d = x; ldx = ld; return GMMIV_OK; }
        void *buf;
        int rc = c->scratch(WS_X, (size_t)T * D * esize(dt), &buf);
        if (rc) return rc;
        GCHK(hipMemcpy2DAsync(buf, D * esize(dt), x, ld * esize(dt), D * esize(dt), T, hipMemcpyHostToDevice, c->stream));
        d = buf; ldx = D;
        return GMMIV_OK;
    }
};

// ---- degenerate inputs (include/gmmiv.h): frames with a non-finite / absurd feature value never reach the kernels ----------
// Screening = one HBM pass over x per call (skipped with the option "assume_finite"); in the -- rare -- call that has unusable
// frames, the usable ones are compacted (k_gather_runs), the entry point runs on them and the per-frame outputs are expanded
// back with the values the rule gives a zero-likelihood frame.  The hot kernels themselves carry no per-element checks.
// Kind (1) frames are not removed from a call: every kernel reads such a value as GMMIV_UNUSABLE_READ_AS (devutil.h, feat_sane), which makes the
// frame a zero-likelihood frame on the device.  What is left for the host side is the COUNT ("screened_frames"): one pass over x that
// flags the frames and adds their number to a device counter -- enqueued, never read back here (option "assume_finite" 1 skips it;
// results do not depend on it).
static int count_unusable(gmmiv_ctx *c, const XView &xv, int dt, int64_t T, int D)
{
    if (c->assume_finite || T <= 0) return GMMIV_OK;
    void *p;
    int rc;
    const size_t fbytes = ((size_t)T + 15) / 16 * 16;
    if ((rc = c->scratch(WS_GFLAG, fbytes + 16, &p))) return rc;
    unsigned char *flag = (unsigned char *)p;
    GCHK(hipMemsetAsync(flag, 0, fbytes + 16, c->stream));
    GCHK(gmmk_flag_frames(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, D, flag, (int *)(flag + fbytes)));
    GCHK(gmmk_count_flags(c->stream, flag, (long)T, c->d_screened));
    return GMMIV_OK;
}

static int check_model(gmmiv_ctx *c, const gmmiv_gmm *g)
{
    if (!c || !g) { gmmiv_set_error("NULL context or model"); return GMMIV_ERR_ARG; }
    if (g->ctx != c) { gmmiv_set_error("model belongs to a different context"); return GMMIV_ERR_ARG; }
    GBIND(c);
    return GMMIV_OK;
}

// ---- frame moments -------------------------------------------------------------------------
int gmmiv_frame_moments(gmmiv_ctx *c, const void *x, int dt, int64_t T, int64_t ldx, int D, double *acc)
{
    if (!c || !acc || T < 0 || D <= 0) { gmmiv_set_error("frame_moments: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    XView xv;
    int rc = xv.init(c, x, dt, T, ldx, D);
    if (rc) return rc;
    DevOut<double> o;
    if ((rc = o.init(c, WS_T0, acc, 2 * D + 1, true))) return rc;
    const int maxb = 2048;
    void *part;
    if ((rc = c->scratch(WS_PART, (size_t)maxb * 2 * D * sizeof(double), &part))) return rc;
    c->t_begin("k_frame_moments");
    GCHK(gmmk_frame_moments(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, D, (double *)part, maxb, o.d));
    c->t_end();
    GCHK(gmmk_add_scalar(c->stream, o.d + 2 * D, (double)T));
    return o.finish();
}

// ---- frame selection -------------------------------------------------------------------------
int gmmiv_gather_frames(gmmiv_ctx *c, const void *x, int dt, int64_t ldx, int D, const int64_t *frame_idx, int64_t n,
                        void *out)
{
    if (!c || !x || !out || !frame_idx || n < 0 || D <= 0 || ldx < D) { gmmiv_set_error("gather_frames: bad argument"); return GMMIV_ERR_ARG; }
    if (!gmmiv_is_device_ptr(x) || !gmmiv_is_device_ptr(out)) { gmmiv_set_error("gather_frames: x and out must be device arrays"); return GMMIV_ERR_ARG; }
    GBIND(c);
    DevIn<int64_t> i_idx;
    int rc = i_idx.init(c, WS_T0, frame_idx, (size_t)n);
    if (rc) return rc;
    c->t_begin("k_gather_frames");
    GCHK(gmmk_gather_frames(c->stream, dt == GMMIV_F64, x, ldx, D, (const long *)i_idx.d, n, out));
    c->t_end();
    if (!gmmiv_is_device_ptr(frame_idx)) GCHK(hipStreamSynchronize(c->stream)); // host index list may be freed
    return GMMIV_OK;
}

int gmmiv_count_unusable_frames(gmmiv_ctx *c, const void *x, int dt, int64_t T, int64_t ldx, int D, int64_t *count)
{
    if (!c || !count || T < 0 || D <= 0) { gmmiv_set_error("count_unusable_frames: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    *count = 0;
    if (T == 0) return GMMIV_OK;
    XView xv;
    int rc = xv.init(c, x, dt, T, ldx, D);
    if (rc) return rc;
    void *p;
    const size_t fbytes = ((size_t)T + 15) / 16 * 16;
    if ((rc = c->scratch(WS_GFLAG, fbytes + 16, &p))) return rc;
    unsigned char *flag = (unsigned char *)p;
    int *any = (int *)(flag + fbytes);
    GCHK(hipMemsetAsync(flag, 0, fbytes + 16, c->stream));
    GCHK(gmmk_flag_frames(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, D, flag, any));
    int h_any = 0;
    GCHK(hipMemcpyAsync(&h_any, any, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    GCHK(hipStreamSynchronize(c->stream));
    if (!h_any) return GMMIV_OK;
    std::vector<unsigned char> hf((size_t)T);
    GCHK(hipMemcpyAsync(hf.data(), flag, (size_t)T, hipMemcpyDeviceToHost, c->stream));
    GCHK(hipStreamSynchronize(c->stream));
    int64_t n = 0;
    for (int64_t t = 0; t < T; ++t) n += hf[t] ? 1 : 0;
    *count = n;
    return GMMIV_OK;
}

int gmmiv_gather_runs(gmmiv_ctx *c, const void *x, int dt, int64_t ldx, int D, const int64_t *runs, int64_t nrun, void *out)
{
    if (!c || !x || !out || (!runs && nrun > 0) || nrun < 0 || D <= 0 || ldx < D) { gmmiv_set_error("gather_runs: bad argument"); return GMMIV_ERR_ARG; }
    if (dt != GMMIV_F32 && dt != GMMIV_F64) { gmmiv_set_error("gather_runs: feature dtype must be GMMIV_F32 or GMMIV_F64"); return GMMIV_ERR_ARG; }
    if (!gmmiv_is_device_ptr(x) || !gmmiv_is_device_ptr(out)) { gmmiv_set_error("gather_runs: x and out must be device arrays"); return GMMIV_ERR_ARG; }
    if (nrun == 0) return GMMIV_OK;
    GBIND(c);
    DevIn<int64_t> i_runs;
    int rc = i_runs.init(c, WS_T0, runs, (size_t)nrun * 3);
    if (rc) return rc;
    c->t_begin("k_gather_runs");
    GCHK(gmmk_gather_runs(c->stream, dt == GMMIV_F64, x, ldx, D, (const long *)i_runs.d, nrun, out));
    c->t_end();
    if (!gmmiv_is_device_ptr(runs)) GCHK(hipStreamSynchronize(c->stream)); // the host table may be freed
    return GMMIV_OK;
}

int gmmiv_segment_means(gmmiv_ctx *c, const double *v, int64_t ld, int nrows, const int64_t *seg_begin, int64_t nseg, double *out)
{
    if (!c || !v || !seg_begin || !out || nrows < 0 || nseg < 0 || ld < 0) { gmmiv_set_error("segment_means: bad argument"); return GMMIV_ERR_ARG; }
    if (!gmmiv_is_device_ptr(v)) { gmmiv_set_error("segment_means: v must be a device array"); return GMMIV_ERR_ARG; }
    if (gmmiv_is_device_ptr(seg_begin)) { gmmiv_set_error("segment_means: seg_begin must be a host array"); return GMMIV_ERR_ARG; }
    for (int64_t s = 0; s < nseg; ++s)
        if (seg_begin[s] < 0 || seg_begin[s + 1] < seg_begin[s]) { gmmiv_set_error("segment_means: seg_begin must be non-negative and non-decreasing"); return GMMIV_ERR_ARG; }
    if (nseg > 0 && nrows > 1 && seg_begin[nseg] > ld) { gmmiv_set_error("segment_means: the last segment ends after the row stride"); return GMMIV_ERR_ARG; }
    const int64_t npair = (int64_t)nrows * nseg;
    if (npair == 0) return GMMIV_OK;
    GBIND(c);
    const int64_t PIECE = 8192;
    std::vector<long> tab; // items [3 x nitem] | pair_off [npair + 1] | pair_len [npair]
    std::vector<long> off(npair + 1, 0), len(npair, 0);
    for (int r = 0; r < nrows; ++r)
        for (int64_t s = 0; s < nseg; ++s) {
            const int64_t p = (int64_t)r * nseg + s;
            len[p] = (long)(seg_begin[s + 1] - seg_begin[s]);
            for (int64_t b = seg_begin[s]; b < seg_begin[s + 1]; b += PIECE) {
                tab.push_back(r); tab.push_back((long)b); tab.push_back((long)(b + PIECE < seg_begin[s + 1] ? b + PIECE : seg_begin[s + 1]));
            }
            off[p + 1] = (long)(tab.size() / 3);
        }
    const size_t nitem = tab.size() / 3;
    tab.insert(tab.end(), off.begin(), off.end());
    tab.insert(tab.end(), len.begin(), len.end());
    void *dtab, *part;
    int rc;
    if ((rc = c->scratch(WS_T8, tab.size() * sizeof(long), &dtab))) return rc;
    if ((rc = c->scratch(WS_T9, (nitem ? nitem : 1) * sizeof(double), &part))) return rc;
    DevOut<double> o;
    if ((rc = o.init(c, WS_T7, out, (size_t)npair, false))) return rc;
    GCHK(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(long), hipMemcpyHostToDevice, c->stream));
    const long *di = (const long *)dtab;
    GCHK(gmmk_segment_means(c->stream, v, (long)ld, di, (long)nitem, (double *)part, di + 3 * nitem, di + 3 * nitem + npair + 1, (long)npair, o.d));
    GCHK(hipStreamSynchronize(c->stream)); // tab is a stack-lifetime vector
    return o.finish();
}

// ---- LLK ---------------------------------------------------------------------------------
// zero-likelihood frames of kind (2) among the n frames whose log-sums K1 has just left in `lse` -> the context's device counter
static int count_dead(gmmiv_ctx *c, const double *lse, int64_t n) { return gmmk_count_dead(c->stream, lse, (long)n, c->d_zero_llk); }

static const void *x_at(const XView &xv, int dt, int64_t frame);
// DETERMINE_TOP_DISTRIBS by the any-shape kernel (k_topc_determine_big), frames in chunks whose logit rows fit 1 GiB of scratch
static int topc_big(gmmiv_ctx *c, const gmmiv_gmm *g, const XView &xv, int dt, int64_t T, int ctop, int complete, double lo, double hi,
                    int *idx, double *lk, double *nlk, double *nllk, double *nw, double *llk)
{
    if (T <= 0) return GMMIV_OK;
    int64_t per = (int64_t)(((size_t)1 << 30) / ((size_t)g->Cp64 * sizeof(double))) / 4 * 4;
    if (per < 4) per = 4;
    if (per > T) per = (T + 3) / 4 * 4;
    void *zs;
    int rc = c->scratch(WS_Z, gmmk_topc_big_scratch_doubles((long)per, g->Cp64) * sizeof(double), &zs);
    if (rc) return rc;
    for (int64_t c0 = 0; c0 < T; c0 += per) { // one timed launch per chunk, like the other chunked paths (kernel_launches counts them)
        const int64_t n = T - c0 < per ? T - c0 : per;
        c->t_begin("k_topc_determine", c0 == 0);
        int krc = gmmk_topc_determine_big(c->stream, dt == GMMIV_F64, x_at(xv, dt, c0), (long)n, xv.ldx, g->D, g->C, g->Cp64, g->meanT, g->ivT, g->lwc,
                                          g->w, ctop, complete, lo, hi, idx ? idx + (size_t)c0 * ctop : nullptr, lk ? lk + (size_t)c0 * ctop : nullptr,
                                          nlk ? nlk + c0 : nullptr, nllk ? nllk + c0 : nullptr, nw ? nw + c0 : nullptr, llk ? llk + c0 : nullptr,
                                          (double *)zs);
        c->t_end();
        if (krc == -1) { gmmiv_set_error("vectSize %d exceeds the generic kernels' bound", g->D); return GMMIV_ERR_UNSUPPORTED; }
        if (krc == -2) { gmmiv_set_error("topDistribsCount %d outside 1 .. mixtureDistribCount %d", ctop, g->C); return GMMIV_ERR_ARG; }
        GCHK(krc);
    }
    return GMMIV_OK;
}

static int run_lse(gmmiv_ctx *c, const gmmiv_gmm *g, const XView &xv, int dt, int64_t T, double **lse_out)
{
    void *lse;
    int rc = c->scratch(WS_LSE, (size_t)(T > 0 ? T : 1) * sizeof(double), &lse);
    if (rc) return rc;
    if (g->KS == GMMK_KS_GENERIC) { // no MFMA instantiation: log sum_c w_c lk_c in the direct form (the top-1 pass of the any-shape kernel, COMPLETE)
        if ((rc = topc_big(c, g, xv, dt, T, 1, 1, -INFINITY, INFINITY, nullptr, nullptr, nullptr, nullptr, nullptr, (double *)lse))) return rc;
        GCHK(count_dead(c, (const double *)lse, T));
        *lse_out = (double *)lse;
        return GMMIV_OK;
    }
    c->t_begin("k_llk_mfma");
    GCHK(gmmk_llk(c->stream, g->KS, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->Pt, g->nct, (double *)lse, (int)(c->use_glds | (c->dbg << 8)), (int)c->wg_waves));
    c->t_end();
    GCHK(count_dead(c, (const double *)lse, T)); // every caller of run_lse is a statistics / likelihood entry point
    *lse_out = (double *)lse;
    return GMMIV_OK;
}

int gmmiv_llk(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, double min_llk,
              double max_llk, double *llk_out, double *sums)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0) { gmmiv_set_error("llk: T < 0"); return GMMIV_ERR_ARG; }
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    double *lse;
    if ((rc = run_lse(c, g, xv, dt, T, &lse))) return rc;
    DevOut<double> o_llk, o_sum;
    if ((rc = o_llk.init(c, WS_T0, llk_out, (size_t)T, false))) return rc;
    if ((rc = o_sum.init(c, WS_T1, sums, 2, true))) return rc;
    void *part;
    if ((rc = c->scratch(WS_SMALL, 3 * 256 * sizeof(double), &part))) return rc;
    GCHK(gmmk_llk_finalize(c->stream, lse, T, min_llk, max_llk, o_llk.d, (double *)part, 1.0, 0.0,
                           sums ? o_sum.d : nullptr, nullptr));
    if (sums) GCHK(gmmk_add_scalar(c->stream, o_sum.d + 1, (double)T));
    if ((rc = o_llk.finish())) return rc;
    return o_sum.finish();
}

// stored-likelihood helpers (defined with the EM path below)
static long z_tile_blocks(int64_t n);
static int64_t z_chunk_frames(gmmiv_ctx *c, const gmmiv_gmm *g);
static const void *x_at(const XView &xv, int dt, int64_t frame);

int gmmiv_llk_determine_top(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, int ctop,
                            int mode, double min_llk, double max_llk, int32_t *idx, double *lk, double *nontop_lk,
                            double *nontop_llk, double *nontop_w, double *llk_out)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || !idx || ctop <= 0) { gmmiv_set_error("determine_top: bad argument"); return GMMIV_ERR_ARG; }
    // no silent clamp: the stride of idx / lk is the caller's ctop, so a caller that asks for more than the model has must
    // clamp on its side (liagpu::computeTestLLR and the Python binding do) -- otherwise its buffers and use_top disagree
    if (ctop > g->C) { gmmiv_set_error("determine_top: topDistribsCount %d exceeds mixtureDistribCount %d (clamp it at the call site)", ctop, g->C); return GMMIV_ERR_ARG; }
    // (topDistribsCount > 64, or a model whose logit rows do not fit LDS -- about 4 700 Gaussians --, go through the any-shape kernel at
    //  the end of this function: mixtureDistribCount / vectSize / topDistribsCount are free keys of the reference, ComputeTest.cpp:129-215)
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    DevOut<int32_t> o_idx;
    DevOut<double> o_lk, o_nlk, o_nllk, o_nw, o_llk;
    if ((rc = o_idx.init(c, WS_T0, idx, (size_t)T * ctop, false))) return rc;
    if ((rc = o_lk.init(c, WS_T1, lk, (size_t)T * ctop, false))) return rc;
    if ((rc = o_nlk.init(c, WS_T2, nontop_lk, (size_t)T, false))) return rc;
    if ((rc = o_nllk.init(c, WS_T3, nontop_llk, (size_t)T, false))) return rc;
    if ((rc = o_nw.init(c, WS_T4, nontop_w, (size_t)T, false))) return rc;
    if ((rc = o_llk.init(c, WS_T5, llk_out, (size_t)T, false))) return rc;
    // Fast path: all logits on the matrix cores (k_llk_mfma<WZ>, the EM path's kernel), selection on the stored likelihoods, the
    // direct form only for the candidates (topc_z.hip).  Redone with the direct-form kernel in the (never observed) case that a
    // non-candidate comes within 1e-6 of the selected set.
    bool done = false;
    // Fused path (default): the candidates are collected in the epilogue of the MFMA log-likelihood kernel itself
    // (k_llk_mfma<TC>: 4 KB of candidate records per frame instead of the 16 KB likelihood round trip) and ranked by
    // k_topc_rank.  Redone by the paths below if a frame overflows its list or fails the margin check (flag).
    if (c->topc_fused && T > 0 && ctop <= 16 && g->C <= 2048 && g->D <= 64 && g->KS <= 15 && c->wg_waves == 8) {
        const size_t per_frame = (size_t)gmmk_topc_cap() * 16;
        size_t budget = (size_t)(c->z_scratch_mb > 0 ? c->z_scratch_mb : 0) << 20;
        if (c->total_mem && budget > c->total_mem / 4) budget = c->total_mem / 4;
        int64_t Tf = (int64_t)(budget / per_frame / 1.2) / 256 * 256;
        if (Tf >= 256) {
            const int64_t first = T < Tf ? (T + 255) / 256 * 256 : Tf;
            const int stats = getenv("GMMIV_TOPC_STATS") != nullptr;
            const int64_t nalloc = first;
            void *cand, *cnt, *th, *sl, *flg;
            if ((rc = c->scratch(WS_Z, (size_t)nalloc * per_frame, &cand))) return rc;
            if ((rc = c->scratch(WS_EIT, (size_t)nalloc * sizeof(int), &cnt))) return rc;
            if ((rc = c->scratch(WS_INV, (size_t)nalloc * (sizeof(double) + sizeof(int)), &sl))) return rc;
            if ((rc = c->scratch(WS_LSE, (size_t)nalloc * sizeof(double), &th))) return rc;
            if ((rc = c->scratch(WS_FLAGS, 64, &flg))) return rc;
            int *efin = (int *)((double *)sl + first);
            void *redo;
            if ((rc = c->scratch(WS_SEG, (size_t)nalloc * 2 * sizeof(long), &redo))) return rc; // second half: the frames k_topc_rank2 hands to k_topc_rank
            int krc = 0;
            bool whole = false; // too many frames failed the fused path: the paths below redo the call
            // one chunk [c0, c0 + n): k_llk_mfma<TC> + k_topc_rank on the context's stream, flags read back, failed frames redone
            auto run_chunk = [&](int64_t c0, int64_t n) -> int {
                GCHK(hipMemsetAsync(flg, 0, 64, c->stream));
                c->t_begin("k_llk_mfma", c0 == 0);
                krc = gmmk_llk_topc(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, (int)c->use_glds,
                                    ctop, (double *)cand, (int *)cnt, (double *)th, (double *)sl, efin);
                c->t_end();
                if (krc) return GMMIV_OK;
                int *oi = o_idx.d + (size_t)c0 * ctop;
                double *olk = o_lk.d ? o_lk.d + (size_t)c0 * ctop : nullptr, *onlk = o_nlk.d ? o_nlk.d + c0 : nullptr;
                double *onllk = o_nllk.d ? o_nllk.d + c0 : nullptr, *onw = o_nw.d ? o_nw.d + c0 : nullptr, *ollk = o_llk.d ? o_llk.d + c0 : nullptr;
                c->t_begin("k_topc_rank", c0 == 0);
                krc = gmmk_topc_rank(c->stream, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->C, (const double *)cand, (const int *)cnt,
                                     (const double *)th, (const double *)sl, efin, g->mean, g->iv, g->lwc, g->w, ctop, mode == GMMIV_TOP_COMPLETE,
                                     min_llk, max_llk, oi, olk, onlk, onllk, onw, ollk, (int *)flg, (long *)redo,
                                     stats | (c->topc_rank_direct ? 2 : 0) | (c->topc_rank2 ? 0 : 4), (long *)redo + nalloc);
                c->t_end();
                if (krc) return GMMIV_OK;
                int hf[16] = {0};
                GCHK(hipMemcpyAsync(hf, flg, 64, hipMemcpyDeviceToHost, c->stream));
                GCHK(hipStreamSynchronize(c->stream));
                if (stats)
                    fprintf(stderr, "topc fused: %ld frames, redone %d | overflow %d, survivors > 64: %d, < ctop: %d, margin %d | list max %d mean %.1f | survivors max %d mean %.1f\n",
                            (long)n, hf[0], hf[1], hf[2], hf[3], hf[4], hf[5], (double)*(unsigned long long *)&hf[6] / (double)n, hf[8],
                            (double)*(unsigned long long *)&hf[10] / (double)n);
                const int64_t nr = hf[0];
                c->topc_fallbacks += nr;
                if (nr == 0) return GMMIV_OK;
                if (nr > n / 8 + 64) { whole = true; return GMMIV_OK; }
                // the few frames whose list overflowed / piled up / failed the margin: direct form for every Gaussian
                // (k_topc_determine) on a gathered copy, results scattered back
                void *gx, *t_idx, *t_d;
                int rc2;
                if ((rc2 = c->scratch(WS_PART, (size_t)nr * g->D * esize(dt), &gx))) return rc2;
                if ((rc2 = c->scratch(WS_T6, (size_t)nr * ctop * sizeof(int), &t_idx))) return rc2;
                if ((rc2 = c->scratch(WS_T7, (size_t)nr * (ctop + 4) * sizeof(double), &t_d))) return rc2;
                double *t_lk = (double *)t_d, *t_nlk = t_lk + (size_t)nr * ctop, *t_nllk = t_nlk + nr, *t_nw = t_nllk + nr, *t_llk = t_nw + nr;
                GCHK(gmmk_gather_frames(c->stream, dt == GMMIV_F64, x_at(xv, dt, c0), xv.ldx, g->D, (const long *)redo, nr, gx));
                c->t_begin("k_topc_determine", c0 == 0);
                GCHK(gmmk_topc_determine(c->stream, dt == GMMIV_F64, gx, nr, g->D, g->D, g->C, g->Cp64, g->meanT, g->ivT, g->lwc, g->w, ctop,
                                         mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, (int *)t_idx, t_lk, t_nlk, t_nllk, t_nw, t_llk));
                c->t_end();
                GCHK(gmmk_topc_scatter(c->stream, nr, ctop, (const long *)redo, (const int *)t_idx, t_lk, t_nlk, t_nllk, t_nw, t_llk, oi, olk, onlk,
                                       onllk, onw, ollk));
                return GMMIV_OK;
            };
            for (int64_t c0 = 0; c0 < T && krc == 0 && !whole; c0 += Tf) {
                const int64_t n = (T - c0) < Tf ? (T - c0) : Tf;
                if ((rc = run_chunk(c0, n))) return rc;
            }
            if (krc > 0) GCHK(krc);
            done = krc == 0 && !whole;
        }
    }
    const int64_t Tc = (!done && c->topc_z) ? z_chunk_frames(c, g) : 0;
    if (Tc > 0 && T > 0 && ctop + 4 <= 64 && gmmk_topc_z_lds(g->nct, g->D)) {
        const int64_t first = T < Tc ? T : Tc;
        const long nfb = z_tile_blocks(first);
        void *zb, *eit, *inv, *lse, *flg;
        if ((rc = c->scratch(WS_Z, (size_t)g->nct * nfb * 2048, &zb))) return rc;
        if ((rc = c->scratch(WS_EIT, (size_t)(g->nct / 2) * nfb * 16 * sizeof(int), &eit))) return rc;
        if ((rc = c->scratch(WS_INV, (size_t)first * (sizeof(double) + sizeof(int)), &inv))) return rc;
        if ((rc = c->scratch(WS_LSE, (size_t)first * sizeof(double), &lse))) return rc;
        if ((rc = c->scratch(WS_FLAGS, 64, &flg))) return rc;
        int *efin = (int *)((double *)inv + first);
        GCHK(hipMemsetAsync(flg, 0, sizeof(int), c->stream));
        for (int64_t c0 = 0; c0 < T; c0 += Tc) {
            const int64_t n = (T - c0) < Tc ? (T - c0) : Tc;
            c->t_begin("k_llk_mfma", c0 == 0);
            GCHK(gmmk_llk_z(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, (double *)lse,
                            (int)c->use_glds, (double *)zb, nfb, (int *)eit, (double *)inv, efin));
            c->t_end();
            c->t_begin("k_topc_from_z", c0 == 0);
            GCHK(gmmk_topc_from_z(c->stream, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->C, g->nct, (const double *)zb, nfb,
                                  (const int *)eit, efin, g->mean, g->iv, g->lwc, g->w, ctop, mode == GMMIV_TOP_COMPLETE, min_llk,
                                  max_llk, o_idx.d + (size_t)c0 * ctop, o_lk.d ? o_lk.d + (size_t)c0 * ctop : nullptr,
                                  o_nlk.d ? o_nlk.d + c0 : nullptr, o_nllk.d ? o_nllk.d + c0 : nullptr, o_nw.d ? o_nw.d + c0 : nullptr,
                                  o_llk.d ? o_llk.d + c0 : nullptr, (int *)flg));
            c->t_end();
        }
        int hflag = 0;
        GCHK(hipMemcpyAsync(&hflag, flg, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
        done = hflag == 0;
    }
    if (!done && (ctop > 64 || !gmmk_topc_frames_per_block(g->Cp64, g->D))) {
        if ((rc = topc_big(c, g, xv, dt, T, ctop, mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, o_idx.d, o_lk.d, o_nlk.d, o_nllk.d, o_nw.d, o_llk.d))) return rc;
    } else if (!done) {
        c->t_begin("k_topc_determine");
        GCHK(gmmk_topc_determine(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->C, g->Cp64, g->meanT, g->ivT, g->lwc,
                                 g->w, ctop, mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, o_idx.d, o_lk.d, o_nlk.d,
                                 o_nllk.d, o_nw.d, o_llk.d));
        c->t_end();
    }
    if ((rc = o_idx.finish())) return rc;
    if ((rc = o_lk.finish())) return rc;
    if ((rc = o_nlk.finish())) return rc;
    if ((rc = o_nllk.finish())) return rc;
    if ((rc = o_nw.finish())) return rc;
    return o_llk.finish();
}

int gmmiv_topgauss_compute(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, int cap, double top_gauss,
                           int mode, double min_llk, double max_llk, int32_t *idx, int32_t *count, double *snsw, double *snsl,
                           double *llk_out, int64_t *n_capped)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || !idx || !count || !snsw || !snsl || cap <= 0 || !(top_gauss > 0.0)) { gmmiv_set_error("topgauss_compute: bad argument"); return GMMIV_ERR_ARG; }
    if (n_capped && gmmiv_is_device_ptr(n_capped)) { gmmiv_set_error("topgauss_compute: n_capped must be a host variable"); return GMMIV_ERR_ARG; }
    const int fixed = top_gauss >= 1.0 ? (int)top_gauss : 0; // _nbg[t] = (unsigned long)topD, TopGauss.cpp:170
    if (fixed > cap) { gmmiv_set_error("topgauss_compute: topGauss %d exceeds the list length %d", fixed, cap); return GMMIV_ERR_ARG; }
    if (n_capped) *n_capped = 0;
    if (T == 0) return GMMIV_OK;
    // the sorted top list of every frame (DETERMINE_TOP_DISTRIBS with topDistribsCount = cap), kept on the device
    void *p;
    const bool idx_dev = gmmiv_is_device_ptr(idx);
    int32_t *d_idx = idx;
    // (slots the DETERMINE pass does not touch: its redo path uses WS_T6 / WS_T7)
    if (!idx_dev) { if ((rc = c->scratch(WS_LP, (size_t)T * cap * sizeof(int32_t), &p))) return rc; d_idx = (int32_t *)p; }
    if ((rc = c->scratch(WS_AUX, (size_t)T * cap * sizeof(double), &p))) return rc;
    double *d_lk = (double *)p;
    const bool llk_dev = llk_out && gmmiv_is_device_ptr(llk_out);
    double *d_llk = llk_out;
    if (!llk_dev) { if ((rc = c->scratch(WS_SLAB, (size_t)T * sizeof(double), &p))) return rc; d_llk = (double *)p; }
    if ((rc = gmmiv_llk_determine_top(c, g, x, dt, T, ldx, cap, mode, min_llk, max_llk, d_idx, d_lk, nullptr, nullptr, nullptr, d_llk))) return rc;
    DevOut<int32_t> o_cnt;
    DevOut<double> o_w, o_l;
    if ((rc = o_cnt.init(c, WS_T0, count, (size_t)T, false))) return rc;
    if ((rc = o_w.init(c, WS_T1, snsw, (size_t)T, false))) return rc;
    if ((rc = o_l.init(c, WS_T2, snsl, (size_t)T, false))) return rc;
    if ((rc = c->scratch(WS_SMALL, 3 * 256 * sizeof(double), &p))) return rc;
    unsigned long long *d_cap = (unsigned long long *)p;
    GCHK(hipMemsetAsync(d_cap, 0, sizeof(unsigned long long), c->stream));
    GCHK(gmmk_topgauss_select(c->stream, T, cap, top_gauss, fixed, g->w, d_idx, d_lk, d_llk, o_cnt.d, o_w.d, o_l.d, d_cap));
    if (!idx_dev) GCHK(hipMemcpyAsync(idx, d_idx, (size_t)T * cap * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (llk_out && !llk_dev) GCHK(hipMemcpyAsync(llk_out, d_llk, (size_t)T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (n_capped) {
        unsigned long long h = 0;
        GCHK(hipMemcpyAsync(&h, d_cap, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
        *n_capped = (int64_t)h;
    } else if (!idx_dev || (llk_out && !llk_dev))
        GCHK(hipStreamSynchronize(c->stream));
    if ((rc = o_cnt.finish())) return rc;
    if ((rc = o_w.finish())) return rc;
    return o_l.finish();
}

int gmmiv_llk_use_top(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, int ctop,
                      const int32_t *idx, const double *nontop_llk, int mode, double min_llk, double max_llk,
                      double *llk_out)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || !idx || !llk_out || ctop <= 0) { gmmiv_set_error("use_top: bad argument"); return GMMIV_ERR_ARG; }
    if (ctop > g->C) { gmmiv_set_error("use_top: topDistribsCount %d exceeds mixtureDistribCount %d", ctop, g->C); return GMMIV_ERR_ARG; }
    if (mode == GMMIV_TOP_COMPLETE && !nontop_llk) { gmmiv_set_error("use_top: COMPLETE mode needs nontop_llk"); return GMMIV_ERR_ARG; }
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    DevIn<int32_t> i_idx;
    DevIn<double> i_n;
    DevOut<double> o_llk;
    if ((rc = i_idx.init(c, WS_T0, idx, (size_t)T * ctop))) return rc;
    if ((rc = i_n.init(c, WS_T1, nontop_llk, (size_t)T))) return rc;
    if ((rc = o_llk.init(c, WS_T2, llk_out, (size_t)T, false))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    c->t_begin("k_topc_use");
    // four lanes per candidate, one frame per wave ("topc_use_lanes" 1: one lane per candidate, four frames per wave) when the
    // selection has at most 16 entries (topc_z.hip); else one wave per frame
    int krc = ctop > 64 ? gmmk_topc_use_big(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->mean, g->iv, g->lwc, g->C, ctop, i_idx.d, i_n.d,
                                            mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, o_llk.d)
              : c->topc_z ? gmmk_topc_use16(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->mean, g->iv, g->lwc, g->C, ctop, i_idx.d, i_n.d,
                                          mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, o_llk.d, (int)c->topc_use_lanes)
                        : -1;
    if (krc == -1 && ctop <= 64)
        krc = gmmk_topc_use(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->mean, g->iv, g->lwc, g->C, ctop, i_idx.d, i_n.d,
                            mode == GMMIV_TOP_COMPLETE, min_llk, max_llk, o_llk.d);
    GCHK(krc);
    c->t_end();
    return o_llk.finish();
}

int gmmiv_llk_use_top_multi(gmmiv_ctx *c, int n_clients, const gmmiv_gmm *const *clients, const void *x, int dt, int64_t T, int64_t ldx,
                            int ctop, const int32_t *idx, const double *nontop_llk, int mode, double min_llk, double max_llk,
                            double *llk_out)
{
    if (!c) { gmmiv_set_error("use_top_multi: null context"); return GMMIV_ERR_ARG; }
    if (n_clients < 0 || (n_clients > 0 && !clients)) { gmmiv_set_error("use_top_multi: bad client list"); return GMMIV_ERR_ARG; }
    if (n_clients == 0) return GMMIV_OK;
    int rc;
    for (int i = 0; i < n_clients; ++i) {
        if ((rc = check_model(c, clients[i]))) return rc;
        if (clients[i]->D != clients[0]->D) { gmmiv_set_error("use_top_multi: client %d has %d dimensions, client 0 has %d", i, clients[i]->D, clients[0]->D); return GMMIV_ERR_ARG; }
        if (ctop > clients[i]->C) { gmmiv_set_error("use_top_multi: topDistribsCount %d exceeds mixtureDistribCount %d of client %d", ctop, clients[i]->C, i); return GMMIV_ERR_ARG; }
    }
    if (T < 0 || !idx || !llk_out || ctop <= 0) { gmmiv_set_error("use_top_multi: bad argument"); return GMMIV_ERR_ARG; }
    if (mode == GMMIV_TOP_COMPLETE && !nontop_llk) { gmmiv_set_error("use_top_multi: COMPLETE mode needs nontop_llk"); return GMMIV_ERR_ARG; }
    const gmmiv_gmm *g0 = clients[0];
    // a HOST output of [n_clients x T] is staged on the device: bound the staging to 1 GiB by going through the clients in groups
    // (a long segment scored against thousands of models would otherwise grow the workspace without limit)
    {
        const size_t cap = (size_t)1 << 30, row = (size_t)(T > 0 ? T : 1) * sizeof(double);
        if (n_clients > 1 && !gmmiv_is_device_ptr(llk_out) && (size_t)n_clients * row > cap) {
            const int per = (int)(cap / row > 0 ? cap / row : 1);
            for (int i0 = 0; i0 < n_clients; i0 += per) {
                const int n = n_clients - i0 < per ? n_clients - i0 : per;
                if ((rc = gmmiv_llk_use_top_multi(c, n, clients + i0, x, dt, T, ldx, ctop, idx, nontop_llk, mode, min_llk, max_llk, llk_out + (size_t)i0 * T))) return rc;
            }
            return GMMIV_OK;
        }
    }
    // one launch for all clients when the four-lanes-per-candidate kernel applies; otherwise client by client (same results either way)
    const bool batched = c->topc_z && c->topc_use_lanes == 4 && ctop <= 16 && g0->D % 2 == 0 && n_clients <= 65535 && T > 0;
    if (!batched) {
        for (int i = 0; i < n_clients; ++i)
            if ((rc = gmmiv_llk_use_top(c, clients[i], x, dt, T, ldx, ctop, idx, nontop_llk, mode, min_llk, max_llk, llk_out + (size_t)i * T))) return rc;
        return GMMIV_OK;
    }
    GBIND(c);
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g0->D))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g0->D))) return rc;
    DevIn<int32_t> i_idx;
    DevIn<double> i_n;
    DevOut<double> o_llk;
    if ((rc = i_idx.init(c, WS_T0, idx, (size_t)T * ctop))) return rc;
    if ((rc = i_n.init(c, WS_T1, nontop_llk, (size_t)T))) return rc;
    if ((rc = o_llk.init(c, WS_T2, llk_out, (size_t)T * n_clients, false))) return rc;
    struct Rec { const double *mean, *iv, *lwc; long C; };
    std::vector<Rec> recs((size_t)n_clients);
    for (int i = 0; i < n_clients; ++i) recs[i] = Rec{clients[i]->mean, clients[i]->iv, clients[i]->lwc, (long)clients[i]->C};
    void *drec;
    if ((rc = c->scratch(WS_T3, recs.size() * sizeof(Rec), &drec))) return rc;
    GCHK(hipMemcpyAsync(drec, recs.data(), recs.size() * sizeof(Rec), hipMemcpyHostToDevice, c->stream));
    GCHK(hipStreamSynchronize(c->stream)); // recs lives on this stack frame
    c->t_begin("k_topc_use");
    GCHK(gmmk_topc_use4_multi(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g0->D, drec, n_clients, ctop, i_idx.d, i_n.d, mode == GMMIV_TOP_COMPLETE,
                              min_llk, max_llk, o_llk.d));
    c->t_end();
    return o_llk.finish();
}

int gmmiv_occ(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, double *gamma)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || !gamma) { gmmiv_set_error("occ: bad argument"); return GMMIV_ERR_ARG; }
    if (T > 0x7fffffff) { gmmiv_set_error("occ: too many frames in one call"); return GMMIV_ERR_UNSUPPORTED; }
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    DevOut<double> o;
    if ((rc = o.init(c, WS_T0, gamma, (size_t)T * g->C, false))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    // Fast path: logits on the matrix cores (k_llk_mfma<WZ>), posteriors = the stored scaled likelihoods rescaled and transposed
    const int64_t Tcz = c->topc_z ? z_chunk_frames(c, g) : 0;
    if (Tcz > 0 && T > 0) {
        const int64_t first = T < Tcz ? T : Tcz;
        const long nfb = z_tile_blocks(first);
        void *zb, *eit, *inv, *lz;
        if ((rc = c->scratch(WS_Z, (size_t)g->nct * nfb * 2048, &zb))) return rc;
        if ((rc = c->scratch(WS_EIT, (size_t)(g->nct / 2) * nfb * 16 * sizeof(int), &eit))) return rc;
        if ((rc = c->scratch(WS_INV, (size_t)first * (sizeof(double) + sizeof(int)), &inv))) return rc;
        if ((rc = c->scratch(WS_LSE, (size_t)first * sizeof(double), &lz))) return rc;
        int *efin = (int *)((double *)inv + first);
        for (int64_t c0 = 0; c0 < T; c0 += Tcz) {
            const int64_t n = (T - c0) < Tcz ? (T - c0) : Tcz;
            c->t_begin("k_llk_mfma", c0 == 0);
            GCHK(gmmk_llk_z(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, (double *)lz,
                            (int)c->use_glds, (double *)zb, nfb, (int *)eit, (double *)inv, efin));
            c->t_end();
            GCHK(count_dead(c, (const double *)lz, n));
            c->t_begin("k_post_from_z", c0 == 0);
            GCHK(gmmk_post_from_z(c->stream, n, g->C, g->nct, (const double *)zb, nfb, (const int *)eit, (const double *)inv, efin,
                                  o.d + (size_t)c0 * g->C));
            c->t_end();
        }
        return o.finish();
    }
    double *lse;
    if ((rc = run_lse(c, g, xv, dt, T, &lse))) return rc;
    c->t_begin("k_posteriors");
    GCHK(gmmk_posteriors(c->stream, dt == GMMIV_F64, xv.d, T, xv.ldx, g->D, g->C, g->Cp64, g->meanT, g->ivT, g->lwc, lse, o.d));
    c->t_end();
    return o.finish();
}

// ---- EM ----------------------------------------------------------------------------------
size_t gmmiv_em_acc_len(int C, int D) { return (size_t)C * (1 + 2 * (size_t)D) + 2; }

// frame segments [0,T) -> nseg pieces aligned to the 64-frame tile; device array in WS_SEG
static int make_chunks(gmmiv_ctx *c, int64_t T, int nseg, long **dev)
{
    const int64_t per = ((T + nseg - 1) / nseg + 63) / 64 * 64;
    void *buf;
    int rc = c->scratch(WS_SEG, (nseg + 1) * sizeof(long), &buf);
    if (rc) return rc;
    GCHK(gmmk_fill_chunks(c->stream, (long *)buf, nseg, (long)per, (long)T)); // seg[i] = min(i per, T), on the device: nothing to wait for
    *dev = (long *)buf;
    return GMMIV_OK;
}

// ---- stored-likelihood path (k_llk_mfma<WZ> + k_stats_z) ----------------------------------------
// Blocks (2 KB) per Gaussian tile of the likelihood scratch for n frames: whole workgroups of the
// log-likelihood kernel (256 frames), then padded so that the tile stride is an ODD number of 4 KB
// granules.  A statistics workgroup reads 16 tiles at the same frame position at once; with a stride
// that is a multiple of the HBM channel interleave (a large power of two) all 16 streams -- and those
// of every other workgroup of the segment -- would sit on the same channel.
static long z_tile_blocks(int64_t n)
{
    long nfb = 16 * ((n + 255) / 256);
    if ((nfb / 2) % 2 == 0) nfb += 2;
    return nfb;
}

// frames per chunk that fit the logit scratch budget (multiple of 64), 0 when the path does not apply
static int64_t z_chunk_frames(gmmiv_ctx *c, const gmmiv_gmm *g)
{
    if (!c->stats_z || g->KS > 15 || c->wg_waves != 8) return 0;
    // The chunk length fixes the segment bounds and with them the fp64 summation order of every reduction of this path, so
    // it depends ONLY on the option and the device's TOTAL memory (the same on every rank of a node), never on what happens
    // to be free: replicated M-steps stay bit-identical.  If the scratch then does not fit, scratch() fails loudly.
    size_t budget = (size_t)(c->z_scratch_mb > 0 ? c->z_scratch_mb : 0) << 20;
    if (c->total_mem && budget > c->total_mem / 4) budget = c->total_mem / 4;
    const size_t per_frame = (size_t)g->nct * 16 * sizeof(double) + (size_t)g->nct * 2 + 16; // likelihoods + exponents
    int64_t tc = (int64_t)(budget / per_frame / 1.2); // scratch() over-allocates by 1/8
    // whole rounds of the log-likelihood kernel: 2 resident workgroups per CU x 256 frames
    const int64_t round = (int64_t)c->n_cu * 2 * 256;
    tc = tc >= round ? tc / round * round : tc / 64 * 64;
    return tc >= 4096 ? tc : 0;
}
static const void *x_at(const XView &xv, int dt, int64_t frame) { return (const char *)xv.d + (size_t)frame * xv.ldx * esize(dt); }

// ---- statistics of a model WITHOUT an MFMA instantiation (vectSize > 80): gamma[t][c] = exp(z_tc - lse_t) by the direct-form VALU
// kernel (k_posteriors), then S[C x NC] (+)= gamma^T [x | 1 | x^2 | 0] on the fp64 GEMM (k_dgemm), frames in chunks whose posterior
// block fits 512 MiB.  S: sum_t g x | sum_t g | sum_t g x^2.  The same sums as MixtureStat::computeAndAccumulateEM
// (AccumulateStat.cpp:103-152) / TVAcc::computeAndAccumulateTVStat (AccumulateTVStat.cpp:332-348), any vectSize.
static int generic_gamma_gemm(gmmiv_ctx *c, const gmmiv_gmm *g, const XView &xv, int dt, int64_t t0, int64_t n, const double *lse, bool sq, int NC,
                              double *S)
{
    int rc;
    int64_t per = (int64_t)(((size_t)512 << 20) / ((size_t)g->C * sizeof(double)));
    if (per < 64) per = 64;
    per = per / 2 * 2; // an even K keeps the aligned GEMM instantiation
    void *gam, *xa;
    const int64_t first = n < per ? n : per;
    if ((rc = c->scratch(WS_Z, (size_t)(first > 0 ? first : 1) * g->C * sizeof(double), &gam))) return rc;
    if ((rc = c->scratch(WS_INV, (size_t)(first > 0 ? first : 1) * NC * sizeof(double), &xa))) return rc;
    if (n <= 0) { GCHK(hipMemsetAsync(S, 0, (size_t)g->C * NC * sizeof(double), c->stream)); return GMMIV_OK; }
    for (int64_t b = 0; b < n; b += per) {
        const int64_t m = n - b < per ? n - b : per;
        c->t_begin("k_posteriors", b == 0);
        GCHK(gmmk_posteriors(c->stream, dt == GMMIV_F64, x_at(xv, dt, t0 + b), (long)m, xv.ldx, g->D, g->C, g->Cp64, g->meanT, g->ivT, g->lwc, lse + t0 + b,
                             (double *)gam));
        c->t_end();
        GCHK(gmmk_build_xa(c->stream, dt == GMMIV_F64, x_at(xv, dt, t0 + b), xv.ldx, g->D, (long)m, sq ? 1 : 0, NC, (double *)xa));
        GCHK(tvk_dgemm(c->stream, true, false, g->C, NC, (int)m, 1.0, (const double *)gam, g->C, 0, (const double *)xa, NC, 0, b == 0 ? 0.0 : 1.0, S, NC, 0, 1));
    }
    return GMMIV_OK;
}

// EM statistics of frames [0, T) into the partial blocks part[nseg] (summed by the caller)
static int em_stats_z(gmmiv_ctx *c, const gmmiv_gmm *g, const XView &xv, int dt, int64_t T, int64_t Tc, double weight,
                      double *lse, int *nseg_out, void **part_out)
{
    int rc;
    const int ngrp = gmmk_stats_z_groups(g->nct);
    int nseg = c->em_chunks > 0 ? (int)c->em_chunks : (c->n_cu * gmmk_stats_z_wg_per_cu() + ngrp - 1) / ngrp;
    nseg = (nseg + 7) / 8 * 8;
    const int64_t first = T < Tc ? T : Tc;
    // at least 256 frames (four tiles) per segment: with 2048 a pass over 12 000 frames (a TrainTarget client) ran 64 workgroups of
    // 24 tiles each -- 0.39 ms where 256 workgroups of 6 tiles take 0.12
    const int64_t cap = (first + 255) / 256;
    if (nseg > cap) nseg = (int)((cap + 7) / 8 * 8);
    const int64_t nchunk = (T + Tc - 1) / Tc;
    // segment bounds relative to the chunk start: one table for full chunks, one for the last chunk
    const int64_t lastn = T - (nchunk - 1) * Tc;
    void *seg, *zb, *part, *eit, *inv;
    if ((rc = c->scratch(WS_SEG, 2 * (size_t)(nseg + 1) * sizeof(long), &seg))) return rc;
    auto fill = [&](long *dst, int64_t n) { // dst[i] = min(i per, n): written on the device, the call does not wait for the stream
        const int64_t per = ((n + nseg - 1) / nseg + 63) / 64 * 64;
        return gmmk_fill_chunks(c->stream, dst, nseg, (long)per, (long)n);
    };
    GCHK(fill((long *)seg, first));
    GCHK(fill((long *)seg + nseg + 1, lastn));
    const long nfb = c->dbg & 64 ? 16 * ((first + 255) / 256) : z_tile_blocks(first);
    if ((rc = c->scratch(WS_Z, (size_t)g->nct * nfb * 2048, &zb))) return rc;
    if ((rc = c->scratch(WS_EIT, (size_t)(g->nct / 2) * nfb * 16 * sizeof(int), &eit))) return rc;
    if ((rc = c->scratch(WS_INV, (size_t)first * (sizeof(double) + sizeof(int)), &inv))) return rc;
    int *efin = (int *)((double *)inv + first);
    const int RL = gmmk_rl_for_ks(g->KS);
    const size_t Cp = (size_t)g->nct * 16;
    if ((rc = c->scratch(WS_PART, (size_t)nseg * Cp * 2 * RL * sizeof(double), &part))) return rc;
    for (int64_t k = 0; k < nchunk; ++k) {
        const int64_t c0 = k * Tc, n = (k == nchunk - 1) ? lastn : Tc;
        c->t_begin("k_llk_mfma", k == 0);
        GCHK(gmmk_llk_z(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, lse + c0,
                        (int)(c->use_glds | ((c->dbg & 15) << 8)), (double *)zb, nfb, (int *)eit, (double *)inv, efin));
        c->t_end();
        GCHK(count_dead(c, lse + c0, n));
        c->t_begin("k_stats_z", k == 0);
        GCHK(gmmk_stats_z(c->stream, g->KS, 1, dt == GMMIV_F64, x_at(xv, dt, c0), xv.ldx, g->D, g->C, g->nct, (const double *)zb, nfb,
                          (const int *)eit, (const double *)inv, efin, weight, (const long *)seg + (k == nchunk - 1 ? nseg + 1 : 0), nseg,
                          (double *)part, nullptr, 0, k > 0, c->prune_thr()));
        c->t_end();
    }
    *nseg_out = nseg;
    *part_out = part;
    return GMMIV_OK;
}

int gmmiv_em_accumulate(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, double weight,
                        double *acc)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || !acc || !(weight > 0.0)) { gmmiv_set_error("em_accumulate: bad argument (T >= 0, weight > 0, acc != NULL)"); return GMMIV_ERR_ARG; }
    const size_t nacc = gmmiv_em_acc_len(g->C, g->D);
    DevOut<double> o;
    if ((rc = o.init(c, WS_T0, acc, nacc, true))) return rc;
    if (T == 0) return o.finish();
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    const int64_t Tc = z_chunk_frames(c, g);
    if (Tc > 0) { // logits written once, statistics from the stored logits
        void *lsew, *small, *part;
        int nseg = 0;
        if ((rc = c->scratch(WS_LSE, (size_t)T * sizeof(double), &lsew))) return rc;
        if ((rc = c->scratch(WS_SMALL, 3 * 256 * sizeof(double), &small))) return rc;
        if ((rc = em_stats_z(c, g, xv, dt, T, Tc, weight, (double *)lsew, &nseg, &part))) return rc;
        GCHK(gmmk_llk_finalize(c->stream, (const double *)lsew, T, -INFINITY, INFINITY, nullptr, (double *)small, 0.0, weight, nullptr,
                               o.d + nacc - 2, weight, o.d + nacc - 1));
        GCHK(gmmk_em_reduce(c->stream, (const double *)part, nseg, g->C, g->nct * 16, g->D, g->KS, o.d));
        return o.finish();
    }
    double *lse;
    if ((rc = run_lse(c, g, xv, dt, T, &lse))) return rc;
    void *small;
    if ((rc = c->scratch(WS_SMALL, 3 * 256 * sizeof(double), &small))) return rc;
    // sum_t weight * log lk_t  and  sum_t weight
    GCHK(gmmk_llk_finalize(c->stream, lse, T, -INFINITY, INFINITY, nullptr, (double *)small, 0.0, weight, nullptr,
                           o.d + nacc - 2, weight, o.d + nacc - 1));
    if (g->KS == GMMK_KS_GENERIC) { // no MFMA instantiation for this vectSize: gamma^T [x | 1 | x^2] on the fp64 GEMM
        const int NC = 2 * g->D + 2;
        void *S;
        if ((rc = c->scratch(WS_PART, (size_t)g->C * NC * sizeof(double), &S))) return rc;
        if ((rc = generic_gamma_gemm(c, g, xv, dt, 0, T, lse, true, NC, (double *)S))) return rc;
        GCHK(gmmk_scatter_em(c->stream, g->C, g->D, NC, (const double *)S, weight, o.d));
        return o.finish();
    }
    // frame chunks: enough workgroups to fill the chip about twice, each >= 4096 frames
    const int nw = c->wg_waves == 8 ? 8 : 4;
    const int ngrp = (g->nct + nw - 1) / nw;
    int nseg = c->em_chunks > 0 ? (int)c->em_chunks : ((nw == 8 ? 2 : 4) * c->n_cu + ngrp - 1) / ngrp;
    nseg = (nseg + 7) / 8 * 8;
    const int64_t cap = (T + 4095) / 4096;
    if (nseg > cap) nseg = (int)cap;
    if (nseg < 1) nseg = 1;
    long *seg;
    if ((rc = make_chunks(c, T, nseg, &seg))) return rc;
    const int RL = gmmk_rl_for_ks(g->KS);
    const size_t Cp = (size_t)g->nct * 16;
    void *part;
    if ((rc = c->scratch(WS_PART, (size_t)nseg * Cp * 2 * RL * sizeof(double), &part))) return rc;
    c->t_begin("k_stats_mfma");
    GCHK(gmmk_stats(c->stream, g->KS, 1, dt == GMMIV_F64, xv.d, xv.ldx, g->D, g->C, g->Pt, g->nct, lse, -log(weight), seg,
                    nseg, (double *)part, nullptr, 0, (int)c->wg_waves, c->prune_arg()));
    c->t_end();
    GCHK(gmmk_em_reduce(c->stream, (const double *)part, nseg, g->C, (int)Cp, g->D, g->KS, o.d));
    return o.finish();
}

int gmmiv_em_get(gmmiv_ctx *c, int C, int D, const double *acc, const double *prev_mean, const double *prev_cov,
                 double *w, double *mean, double *cov)
{
    if (!c || !acc || !prev_mean || !prev_cov || !w || !mean || !cov || C <= 0 || D <= 0) { gmmiv_set_error("em_get: bad argument"); return GMMIV_ERR_ARG; }
    GBIND(c);
    const size_t CD = (size_t)C * D;
    DevIn<double> i_acc, i_pm, i_pc;
    DevOut<double> o_w, o_m, o_c;
    int rc;
    if ((rc = i_acc.init(c, WS_T0, acc, gmmiv_em_acc_len(C, D)))) return rc;
    if ((rc = i_pm.init(c, WS_T1, prev_mean, CD))) return rc;
    if ((rc = i_pc.init(c, WS_T2, prev_cov, CD))) return rc;
    if ((rc = o_w.init(c, WS_T3, w, C, false))) return rc;
    if ((rc = o_m.init(c, WS_T4, mean, CD, false))) return rc;
    if ((rc = o_c.init(c, WS_T5, cov, CD, false))) return rc;
    GCHK(gmmk_em_get(c->stream, C, D, i_acc.d, i_pm.d, i_pc.d, o_w.d, o_m.d, o_c.d));
    if ((rc = o_w.finish())) return rc;
    if ((rc = o_m.finish())) return rc;
    return o_c.finish();
}

int gmmiv_variance_control(gmmiv_ctx *c, int C, int D, double *cov, double flooring, double ceiling,
                           const double *cov_signal, int64_t *counts)
{
    if (!c || C <= 0 || D <= 0 || !cov || !cov_signal) { gmmiv_set_error("variance_control: bad argument"); return GMMIV_ERR_ARG; }
    if (counts && gmmiv_is_device_ptr(counts)) { gmmiv_set_error("variance_control: counts must be a host array"); return GMMIV_ERR_ARG; }
    GBIND(c);
    DevOut<double> o;
    DevIn<double> i_cs;
    int rc;
    if ((rc = o.init(c, WS_T0, cov, (size_t)C * D, true))) return rc;
    if ((rc = i_cs.init(c, WS_T1, cov_signal, D))) return rc;
    unsigned long long *dc = nullptr;
    if (counts) {
        void *p;
        if ((rc = c->scratch(WS_SMALL, 3 * 256 * sizeof(double), &p))) return rc;
        dc = (unsigned long long *)p;
        GCHK(hipMemsetAsync(dc, 0, 2 * sizeof(unsigned long long), c->stream));
    }
    GCHK(gmmk_variance_control(c->stream, C, D, o.d, flooring, ceiling, i_cs.d, dc));
    if (counts) {
        unsigned long long h[2];
        GCHK(hipMemcpyAsync(h, dc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        GCHK(hipStreamSynchronize(c->stream));
        counts[0] = (int64_t)h[0];
        counts[1] = (int64_t)h[1];
    }
    return o.finish();
}

// ---- Baum-Welch N / F ----------------------------------------------------------------------
int gmmiv_tv_stats(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx,
                   const int64_t *utt_begin, int64_t U, double *N, double *F)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (T < 0 || U < 0 || !utt_begin || !N || !F) { gmmiv_set_error("tv_stats: bad argument"); return GMMIV_ERR_ARG; }
    if (gmmiv_is_device_ptr(utt_begin)) { gmmiv_set_error("tv_stats: utt_begin must be a host array"); return GMMIV_ERR_ARG; }
    if (U == 0) return GMMIV_OK;
    if (utt_begin[0] < 0 || utt_begin[U] > T) { gmmiv_set_error("tv_stats: utt_begin out of range"); return GMMIV_ERR_ARG; }
    for (int64_t u = 0; u < U; ++u)
        if (utt_begin[u + 1] < utt_begin[u]) { gmmiv_set_error("tv_stats: utt_begin must be non-decreasing"); return GMMIV_ERR_ARG; }
    if (U > 0x7fffffff / 64) { gmmiv_set_error("tv_stats: too many utterances in one call"); return GMMIV_ERR_UNSUPPORTED; }
    XView xv;
    if ((rc = xv.init(c, x, dt, T, ldx, g->D))) return rc;
    if ((rc = count_unusable(c, xv, dt, T, g->D))) return rc;
    std::vector<long> h(utt_begin, utt_begin + U + 1);
    void *seg;
    if ((rc = c->scratch(WS_SEG, (U + 1) * sizeof(long), &seg))) return rc;
    GCHK(hipMemcpyAsync(seg, h.data(), (U + 1) * sizeof(long), hipMemcpyHostToDevice, c->stream));
    GCHK(hipStreamSynchronize(c->stream));
    const size_t SV = (size_t)g->C * g->D;
    DevOut<double> o_n, o_f;
    if ((rc = o_n.init(c, WS_T0, N, (size_t)U * g->C, false))) return rc;
    if ((rc = o_f.init(c, WS_T1, F, (size_t)U * SV, false))) return rc;
    const int64_t Tcz = z_chunk_frames(c, g);
    if (Tcz > 0) {
        // chunks of whole utterances whose frames fit the logit scratch; an utterance longer than
        // that sends the call to the recomputing kernel below
        std::vector<int64_t> cu(1, 0); // chunk k = utterances [cu[k], cu[k+1])
        bool fits = true;
        int64_t maxn = 0;
        // chunks of about EQUAL length (the last one of a greedy fill is a fraction of the others and runs the chip half empty:
        // 512 utterances x 3000 frames on a 0.79 M-frame scratch went 283 + 229, 3 % slower than 256 + 256)
        const int64_t Ttot = utt_begin[U] - utt_begin[0], nch = (Ttot + Tcz - 1) / Tcz;
        const int64_t target = nch > 0 ? (Ttot + nch - 1) / nch : Tcz;
        for (int64_t u = 0; u < U;) {
            int64_t v = u;
            while (v < U && utt_begin[v + 1] - utt_begin[u] <= Tcz && (v == u || utt_begin[v] - utt_begin[u] < target)) ++v;
            // whole rounds of the statistics kernel: it runs one workgroup per (utterance, Gaussian group), 64 utterances fill the chip
            // an integer number of times in every shape -- 262 utterances of 3000 frames took 9 rounds where 256 take 8
            if (v < U && v - u > 64) v = u + (v - u) / 64 * 64;
            if (v == u) { fits = false; break; }
            if (utt_begin[v] - utt_begin[u] > maxn) maxn = utt_begin[v] - utt_begin[u];
            cu.push_back(v);
            u = v;
        }
        if (fits) {
            // relative segment bounds, chunk after chunk: chunk k occupies cu[k] + k .. cu[k+1] + k
            std::vector<long> rel;
            for (size_t k = 0; k + 1 < cu.size(); ++k)
                for (int64_t u = cu[k]; u <= cu[k + 1]; ++u) rel.push_back((long)(utt_begin[u] - utt_begin[cu[k]]));
            void *zb, *lsew, *eit, *inv;
            if ((rc = c->scratch(WS_SEG, rel.size() * sizeof(long), &seg))) return rc;
            GCHK(hipMemcpyAsync(seg, rel.data(), rel.size() * sizeof(long), hipMemcpyHostToDevice, c->stream));
            GCHK(hipStreamSynchronize(c->stream));
            const long nfb = z_tile_blocks(maxn);
            if ((rc = c->scratch(WS_Z, (size_t)g->nct * nfb * 2048, &zb))) return rc;
            if ((rc = c->scratch(WS_LSE, (size_t)(maxn > 0 ? maxn : 1) * sizeof(double), &lsew))) return rc;
            if ((rc = c->scratch(WS_EIT, (size_t)(g->nct / 2) * nfb * 16 * sizeof(int), &eit))) return rc;
            if ((rc = c->scratch(WS_INV, (size_t)(maxn > 0 ? maxn : 1) * (sizeof(double) + sizeof(int)), &inv))) return rc;
            int *efin = (int *)((double *)inv + (maxn > 0 ? maxn : 1));
            // A few utterances (on-line extraction: ONE) give the statistics kernel a few workgroups -- 8 per utterance, each walking all
            // its frames: 0.39 ms for 3000 frames.  Up to 16 utterances in one chunk are cut into about 32 pieces of whole 64-frame
            // tiles; the pieces are "utterances" of the kernel, written to scratch rows and summed back in piece order.
            if (cu.size() == 2 && U <= 16 && c->tv_stats_split) {
                std::vector<long> pb;
                std::vector<int> rbh((size_t)U + 1, 0);
                const int64_t per_utt = 32 / U > 1 ? 32 / U : 1;
                for (int64_t u = 0; u < U; ++u) {
                    const int64_t b = utt_begin[u] - utt_begin[0], e = utt_begin[u + 1] - utt_begin[0], len = e - b;
                    int64_t S = (len + 255) / 256;
                    S = S < 1 ? 1 : (S > per_utt ? per_utt : S);
                    const int64_t per = ((len + S - 1) / S + 63) / 64 * 64;
                    int64_t pos = b;
                    int cntp = 0;
                    do { pb.push_back((long)pos); pos = per > 0 && pos + per < e ? pos + per : e; ++cntp; } while (pos < e);
                    rbh[u + 1] = rbh[u] + cntp;
                }
                pb.push_back((long)(utt_begin[U] - utt_begin[0]));
                const int np = rbh[U];
                if (np > U) {
                    void *pseg, *prb, *tn, *tf;
                    if ((rc = c->scratch(WS_SEG, pb.size() * sizeof(long), &pseg))) return rc;
                    if ((rc = c->scratch(WS_SMALL, rbh.size() * sizeof(int) + 64, &prb))) return rc;
                    if ((rc = c->scratch(WS_T2, (size_t)np * g->C * sizeof(double), &tn))) return rc;
                    if ((rc = c->scratch(WS_T3, (size_t)np * SV * sizeof(double), &tf))) return rc;
                    GCHK(hipMemcpyAsync(pseg, pb.data(), pb.size() * sizeof(long), hipMemcpyHostToDevice, c->stream));
                    GCHK(hipMemcpyAsync(prb, rbh.data(), rbh.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                    GCHK(hipStreamSynchronize(c->stream)); // pb / rbh live on this stack frame
                    const int64_t c0 = utt_begin[0], n = utt_begin[U] - c0;
                    c->t_begin("k_llk_mfma", true);
                    GCHK(gmmk_llk_z(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, (double *)lsew,
                                    (int)(c->use_glds | ((c->dbg & 15) << 8)), (double *)zb, nfb, (int *)eit, (double *)inv, efin));
                    c->t_end();
                    GCHK(count_dead(c, (const double *)lsew, n));
                    c->t_begin("k_stats_z", true);
                    GCHK(gmmk_stats_z(c->stream, g->KS, 0, dt == GMMIV_F64, x_at(xv, dt, c0), xv.ldx, g->D, g->C, g->nct, (const double *)zb, nfb,
                                      (const int *)eit, (const double *)inv, efin, 1.0, (const long *)pseg, np, (double *)tn, (double *)tf, 1, 0,
                                      c->prune_thr()));
                    GCHK(gmmk_rows_sum_groups(c->stream, g->C, (int)U, (const int *)prb, (const double *)tn, o_n.d));
                    GCHK(gmmk_rows_sum_groups(c->stream, (long)SV, (int)U, (const int *)prb, (const double *)tf, o_f.d));
                    c->t_end();
                    if ((rc = o_n.finish())) return rc;
                    return o_f.finish();
                }
            }
            for (size_t k = 0; k + 1 < cu.size(); ++k) {
                const int64_t u0 = cu[k], u1 = cu[k + 1], c0 = utt_begin[u0], n = utt_begin[u1] - c0;
                c->t_begin("k_llk_mfma", k == 0);
                GCHK(gmmk_llk_z(c->stream, g->KS, dt == GMMIV_F64, x_at(xv, dt, c0), n, xv.ldx, g->D, g->Pt, g->nct, (double *)lsew,
                                (int)(c->use_glds | ((c->dbg & 15) << 8)), (double *)zb, nfb, (int *)eit, (double *)inv, efin));
                c->t_end();
                GCHK(count_dead(c, (const double *)lsew, n));
                c->t_begin("k_stats_z", k == 0);
                GCHK(gmmk_stats_z(c->stream, g->KS, 0, dt == GMMIV_F64, x_at(xv, dt, c0), xv.ldx, g->D, g->C, g->nct, (const double *)zb,
                                  nfb, (const int *)eit, (const double *)inv, efin, 1.0, (const long *)seg + u0 + k, (int)(u1 - u0),
                                  o_n.d + (size_t)u0 * g->C, o_f.d + (size_t)u0 * SV, 1, 0, c->prune_thr()));
                c->t_end();
            }
            if ((rc = o_n.finish())) return rc;
            return o_f.finish();
        }
    }
    double *lse;
    if ((rc = run_lse(c, g, xv, dt, T, &lse))) return rc;
    if (g->KS == GMMK_KS_GENERIC) { // utterance by utterance: S_u = gamma_u^T [x | 1], N row = its last column, F row = the rest
        const int NC = g->D + 2 - (g->D & 1); // [x | 1] padded to an even width
        void *S;
        if ((rc = c->scratch(WS_PART, (size_t)g->C * NC * sizeof(double), &S))) return rc;
        for (int64_t u = 0; u < U; ++u) {
            if ((rc = generic_gamma_gemm(c, g, xv, dt, utt_begin[u], utt_begin[u + 1] - utt_begin[u], lse, false, NC, (double *)S))) return rc;
            GCHK(gmmk_scatter_nf(c->stream, g->C, g->D, NC, (const double *)S, o_n.d + (size_t)u * g->C, o_f.d + (size_t)u * SV));
        }
        if ((rc = o_n.finish())) return rc;
        return o_f.finish();
    }
    // every (u, c < C) row is written by exactly one wave (zeros for an empty utterance)
    c->t_begin("k_stats_mfma");
    GCHK(gmmk_stats(c->stream, g->KS, 0, dt == GMMIV_F64, xv.d, xv.ldx, g->D, g->C, g->Pt, g->nct, lse, 0.0,
                    (const long *)seg, (int)U, o_n.d, o_f.d, 1, (int)c->wg_waves, c->prune_arg()));
    c->t_end();
    if ((rc = o_n.finish())) return rc;
    return o_f.finish();
}

// TVAcc::computeAndAccumulateTVStat with the reference's file -> ndx-line map (AccumulateTVStat.cpp:318-346: the statistics of a
// feature file are added to EVERY line `locidcs` that lists it, and a line sums all of its files): the Baum-Welch pass runs
// ONCE per file, the line rows are sums of file rows.
int gmmiv_tv_stats_lines(gmmiv_ctx *c, const gmmiv_gmm *g, const void *x, int dt, int64_t T, int64_t ldx, const int64_t *file_begin,
                         int64_t nfiles, int64_t nlines, const int64_t *line_off, const int64_t *line_files, double *N, double *F)
{
    int rc = check_model(c, g);
    if (rc) return rc;
    if (nfiles < 0 || nlines < 0 || !file_begin || !line_off || (!line_files && nlines && line_off[nlines]) || !N || !F) { gmmiv_set_error("tv_stats_lines: bad argument"); return GMMIV_ERR_ARG; }
    if (gmmiv_is_device_ptr(line_off) || gmmiv_is_device_ptr(line_files)) { gmmiv_set_error("tv_stats_lines: the line map must be host arrays"); return GMMIV_ERR_ARG; }
    if (nlines == 0) return GMMIV_OK;
    if (line_off[0] != 0) { gmmiv_set_error("tv_stats_lines: line_off must start at 0"); return GMMIV_ERR_ARG; }
    for (int64_t l = 0; l < nlines; ++l)
        if (line_off[l + 1] < line_off[l]) { gmmiv_set_error("tv_stats_lines: line_off must be non-decreasing"); return GMMIV_ERR_ARG; }
    const int64_t npairs = line_off[nlines];
    for (int64_t p = 0; p < npairs; ++p)
        if (line_files[p] < 0 || line_files[p] >= nfiles) { gmmiv_set_error("tv_stats_lines: file index %lld out of range", (long long)line_files[p]); return GMMIV_ERR_ARG; }
    const size_t SV = (size_t)g->C * g->D;
    void *pn, *pf, *pm;
    if ((rc = c->scratch(WS_T6, (size_t)(nfiles ? nfiles : 1) * g->C * sizeof(double), &pn))) return rc;
    if ((rc = c->scratch(WS_T7, (size_t)(nfiles ? nfiles : 1) * SV * sizeof(double), &pf))) return rc;
    if ((rc = gmmiv_tv_stats(c, g, x, dt, T, ldx, file_begin, nfiles, (double *)pn, (double *)pf))) return rc;   // once per file, on the device
    if ((rc = c->scratch(WS_T8, (size_t)(nlines + 1 + npairs + 1) * sizeof(long), &pm))) return rc;
    std::vector<long> h((size_t)nlines + 1 + npairs);
    for (int64_t l = 0; l <= nlines; ++l) h[l] = (long)line_off[l];
    for (int64_t p = 0; p < npairs; ++p) h[nlines + 1 + p] = (long)line_files[p];
    GCHK(hipMemcpyAsync(pm, h.data(), h.size() * sizeof(long), hipMemcpyHostToDevice, c->stream));
    DevOut<double> o_n, o_f;
    if ((rc = o_n.init(c, WS_T4, N, (size_t)nlines * g->C, false))) return rc;
    if ((rc = o_f.init(c, WS_T5, F, (size_t)nlines * SV, false))) return rc;
    const long *off = (const long *)pm, *rows = off + nlines + 1;
    GCHK(tvk_merge_rows(c->stream, (long)nlines, g->C, off, rows, (const double *)pn, o_n.d));
    GCHK(tvk_merge_rows(c->stream, (long)nlines, (long)SV, off, rows, (const double *)pf, o_f.d));
    GCHK(hipStreamSynchronize(c->stream)); // h is a stack-lifetime vector
    if ((rc = o_n.finish())) return rc;
    return o_f.finish();
}


} // extern "C"
