// ctx.h -- context, workspace and host/device pointer plumbing behind include/gmmiv.h (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/gmmiv.h"

void gmmiv_set_error(const char *fmt, ...);

#include "kopts.h"

#define GBIND(c)                                                                                    \
    do {                                                                                            \
        GCHK(hipSetDevice((c)->device));                                                            \
        gmmiv_kopts_bind(&(c)->ko);                                                                 \
    } while (0)

#define GCHK(expr)                                                                                  \
    do {                                                                                            \
        hipError_t _e = (hipError_t)(expr);                                                         \
        if (_e != hipSuccess) {                                                                     \
            gmmiv_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
            return GMMIV_ERR_HIP;                                                                   \
        }                                                                                           \
    } while (0)

enum { WS_X = 0, WS_LSE, WS_PART, WS_SEG, WS_SMALL, WS_T0, WS_T1, WS_T2, WS_T3, WS_T4, WS_T5, WS_T6, WS_T7, WS_T8,
       WS_T9, WS_TIV, WS_LP, WS_AUX, WS_SLAB, WS_SLOTS, WS_FLAGS, WS_Z, WS_EIT, WS_INV,
       WS_GFLAG,
       WS_COUNT }; // WS_GFLAG: the per-frame flags of the kind-(1) counting pass

struct gmmiv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void *ws[WS_COUNT] = {};
    size_t ws_size[WS_COUNT] = {};
    // options
    long use_glds = 1;
    long em_chunks = 0; // 0 = auto
    long timing = 0;
    long dbg = 0; // timing experiments (wrong results when != 0)
    // statistics kernels, OPT-IN: groups of 4 frames x 16 Gaussians whose posteriors are ALL below
    // 2^-prune_log2 are skipped (exp + MFMAs).  0 (default) = never skip: every pair is accumulated
    // like the reference does.  With 100, at most 2e10 pairs x 2^-100 = 1.6e-20 of posterior mass is
    // dropped per 10 M-frame pass -- invisible in any live Gaussian, but a DEAD Gaussian (occupancy
    // ~1e-60) gets a different ML mean than the reference's sum of denormal-scale terms.
    long prune_log2 = 0;
    double prune_arg() const { return prune_log2 > 0 ? -(double)prune_log2 * 0.6931471805599453 : -__builtin_inf(); }
    double prune_thr() const { return prune_log2 > 0 ? __builtin_ldexp(1.0, -(int)prune_log2) : 0.0; } // as a posterior
    long wg_waves = 8; // waves per workgroup of the two MFMA GMM kernels (8, or 4 for A/B runs)
    // 1 (default): the log-likelihood kernel leaves the logits in HBM and the statistics kernel reads
    // them back (stats_z.hip) instead of recomputing them; 0: the recomputing k_stats_mfma
    long stats_z = 1;
    long topc_use_lanes = 4;   // USE_TOP_DISTRIBS: lanes per candidate -- 4 (k_topc_use4, one frame per wave) or 1 (k_topc_use16, four frames per wave)
    long topc_rank2 = 1;       // k_topc_rank2 (two frames per wave) + k_topc_rank on the frames it passes on; 0 = k_topc_rank for every frame
    long topc_rank_direct = 0; // k_topc_rank: 1 = every frame's survivors re-evaluated in the direct form (round 2); 0 = only near-ties
    long topc_fused = 1; // DETERMINE_TOP_DISTRIBS with the candidates collected inside k_llk_mfma<TC> (no likelihood round trip); 0: topc_z
    // 1: the caller vouches that every feature value is finite and |x| <= 1e18 -- the counting pass over the features (one read of x
    // per call, enqueued: no synchronisation) is skipped.  RESULTS do not depend on it: every kernel reads an unusable value as
    // GMMIV_UNUSABLE_READ_AS (devutil.h), which makes its frame a zero-likelihood frame on the device.
    long assume_finite = 0;
    // the rule's two counters live ON THE DEVICE (no host synchronisation when they are bumped; options "zero_llk_frames" /
    // "screened_frames" read them): frames whose likelihood under the call's model is 0 in fp64 (k_count_dead after the log-likelihood
    // kernel of gmmiv_llk / _em_accumulate / _tv_stats(_lines) / _occ; kind-(1) frames are among them) and frames with an unusable
    // feature value (k_flag_frames + k_count_flags at the top of every frame-consuming call)
    unsigned long long *d_zero_llk = nullptr, *d_screened = nullptr;
    long topc_fallbacks = 0; // calls the fused path handed to the slower paths (list overflow / margin check); read with set_option
    long topc_z = 1;     // DETERMINE_TOP_DISTRIBS from the stored MFMA likelihoods (topc_z.hip); 0: the direct-form VALU kernel
    long tv_acc_mb = 8192; // T-matrix E-step: MiB of packed E_u kept per super-batch before A / Cmx are updated (one GEMM with K = its utterances)
    long tv_md_device = 1; // minDivergence: R normalised and factored on the device (one workgroup of k_chol_left); 0: on the host
    long tv_mstep_solve = 1; // updateTestimate by substitution through the Cholesky factor (k_chol_solve_multi); 0: explicit inverse + GEMM
    long tv_stats_split = 1; // gmmiv_tv_stats on at most 16 utterances: each utterance in pieces of whole tiles (more workgroups), summed back; 0 = one segment per utterance
    long tv_tett_direct = 1; // estimateTETt by k_tett_packed (lower triangle only, written packed); 0 = batched GEMM + pack
    long tv_batch = 1024; // utterances per batch of the i-vector solve / T-matrix E-step (one workgroup per system)
    gmmiv_kopts ko;   // "z_waves", "z_tv4", "z_depth_*", "gemm_*", "chol_*": see gmmiv_kopts above
    // host callbacks at the two points of a T-matrix EM iteration where a collective can start early (gmmiv_ctx_set_hook)
    struct Hook { gmmiv_hook_fn fn = nullptr; void *user = nullptr; void call() const { if (fn) fn(user); } };
    Hook hook_tv_a_ready, hook_md_factored;
    // logit scratch budget (MiB): frames are processed in chunks that fit.  16 GiB = 0.85 M frames of a 2048-Gaussian model per
    // chunk.  Measured on 10 M frames (tools/alloc_time.py): 8 GiB 125.3, 16 GiB 126.4, 32 GiB 126.8, 64 GiB 126.9 G pairs/s -- but a
    // large hipMalloc can take seconds: VRAM that earlier allocations (of this or of an earlier process) have dirtied is cleared at ~26 GB/s
    // when it is handed out again -- once the driver's clean pool (about 60 GB on a fresh box, tools/malloc_probe_torch.py) is used up:
    // 1.1 s for 30 GB, 2.3 s for 60 GB on the FIRST call of a context in a process that had held other buffers, against 0.25 ms for
    // 14 GB.  Rounds 1-3 used 64 GiB: 0.4 % more throughput for up to 2.3 s and 46 GB.
    long z_scratch_mb = 16384;
    int n_cu = 256;
    // gmmiv_score_plda: K_n = (n FTJF + I)^-1 and log det K_n per session count n, kept while FTJF stays the same matrix
    // (each costs an O(rankF^3) inverse on the host: 3 ms per call at rankF = 200 when recomputed every time)
    std::vector<double> plda_ftjf;
    struct PldaK { std::vector<double> K; double alpha; };
    std::map<long, PldaK> plda_k;
    // communicators created on this context (capi_comm.hip): released with the context if the caller has not done so
    std::vector<struct gmmiv_comm *> comms;
    size_t total_mem = 0; // device memory size (bounds the likelihood scratch deterministically)
    // HIP-event timing of the kernels of the last call (option "timing"): one slot per kernel name,
    // one event pair per launch of that kernel inside the call
    enum { NSLOT = 6 };
    std::vector<hipEvent_t> ev0[NSLOT], ev1[NSLOT];
    const char *ev_name[NSLOT] = {};
    int ev_used[NSLOT] = {};
    int ev_cur = -1, ev_last = -1;

    // grow-only device scratch; contents are NOT preserved across a growth
    int scratch(int slot, size_t bytes, void **out)
    {
        if (bytes == 0) bytes = 8;
        if (ws_size[slot] < bytes) {
            if (ws[slot]) {
                GCHK(hipStreamSynchronize(stream));
                GCHK(hipFree(ws[slot]));
                ws[slot] = nullptr;
                ws_size[slot] = 0;
            }
            size_t want = bytes + bytes / 8;
            static const bool trace = [] { const char *e = getenv("GMMIV_TRACE_ALLOC"); return e && *e && *e != '0'; }();
            timespec t0, t1;
            if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
            hipError_t me = hipMalloc(&ws[slot], want);
            if (trace) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                fprintf(stderr, "[gmmiv] workspace %d: hipMalloc of %.1f MiB took %.2f ms\n", slot, want / 1048576.0,
                        (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
            }
            if (me != hipSuccess) {
                (void)hipGetLastError();
                ws[slot] = nullptr;
                gmmiv_set_error("device workspace %d: hipMalloc of %zu MiB failed (%s); lower the \"z_scratch_mb\" / \"tv_batch\" options or free device memory",
                                slot, want >> 20, hipGetErrorString(me));
                return GMMIV_ERR_HIP;
            }
            ws_size[slot] = want;
        }
        *out = ws[slot];
        return GMMIV_OK;
    }
    // first = true starts a new measurement of `name` (first launch of an API call), false adds a launch
    void t_begin(const char *name, bool first = true)
    {
        if (!timing) return;
        int s = -1;
        for (int i = 0; i < NSLOT; ++i)
            if (ev_name[i] && !strcmp(ev_name[i], name)) { s = i; break; }
        if (s < 0)
            for (int i = 0; i < NSLOT; ++i)
                if (!ev_name[i]) { s = i; break; }
        if (s < 0) s = NSLOT - 1;
        if (ev_name[s] == nullptr || strcmp(ev_name[s], name) != 0 || first) ev_used[s] = 0;
        ev_name[s] = name;
        if ((int)ev0[s].size() <= ev_used[s]) {
            hipEvent_t a = nullptr, b = nullptr;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            ev0[s].push_back(a); ev1[s].push_back(b);
        }
        ev_cur = s;
        (void)hipEventRecord(ev0[s][ev_used[s]], stream);
    }
    void t_end()
    {
        if (!timing || ev_cur < 0) return;
        (void)hipEventRecord(ev1[ev_cur][ev_used[ev_cur]], stream);
        ev_used[ev_cur]++;
        ev_last = ev_cur;
        ev_cur = -1;
    }
    // total milliseconds over the launches of the last call
    double t_query(int s)
    {
        if (s < 0 || s >= NSLOT || ev_used[s] <= 0) return -1.0;
        double tot = 0.0;
        for (int i = 0; i < ev_used[s]; ++i) {
            if (hipEventSynchronize(ev1[s][i]) != hipSuccess) return -1.0;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev0[s][i], ev1[s][i]) != hipSuccess) return -1.0;
            tot += (double)ms;
        }
        return tot;
    }
};

bool gmmiv_is_device_ptr(const void *p);
void gmmiv_comm_orphan(struct gmmiv_comm *comm); // capi_comm.hip: the context of `comm` is going away

// Read-only argument: device view of a host-or-device array (copies host data into a scratch slot).
template <typename T> struct DevIn {
    const T *d = nullptr;
    int init(gmmiv_ctx *c, int slot, const T *p, size_t n)
    {
        if (!p || n == 0) { d = p; return GMMIV_OK; }
        if (gmmiv_is_device_ptr(p)) { d = p; return GMMIV_OK; }
        void *buf;
        int rc = c->scratch(slot, n * sizeof(T), &buf);
        if (rc) return rc;
        GCHK(hipMemcpyAsync(buf, p, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
        d = (const T *)buf;
        return GMMIV_OK;
    }
};

// Output (optionally read-modify-write) argument.
template <typename T> struct DevOut {
    T *d = nullptr;
    T *host = nullptr;
    size_t n = 0;
    gmmiv_ctx *c = nullptr;
    int init(gmmiv_ctx *ctx, int slot, T *p, size_t count, bool load)
    {
        c = ctx; n = count;
        if (!p || count == 0) { d = p; return GMMIV_OK; }
        if (gmmiv_is_device_ptr(p)) { d = p; return GMMIV_OK; }
        void *buf;
        int rc = ctx->scratch(slot, count * sizeof(T), &buf);
        if (rc) return rc;
        d = (T *)buf;
        host = p;
        if (load) GCHK(hipMemcpyAsync(buf, p, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return GMMIV_OK;
    }
    int finish()
    {
        if (host) {
            GCHK(hipMemcpyAsync(host, d, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
            GCHK(hipStreamSynchronize(c->stream));
        }
        return GMMIV_OK;
    }
};

struct gmmiv_gmm {
    gmmiv_ctx *ctx = nullptr;
    int C = 0, D = 0, KS = 0, nct = 0, Cp64 = 0;
    double *w = nullptr, *mean = nullptr, *iv = nullptr; // row-major copies
    double *a = nullptr, *lwc = nullptr;                 // a_c, log(w cst) (padded)
    double *Pt = nullptr;                                // MFMA-ordered operands
    double *meanT = nullptr, *ivT = nullptr;             // [D][Cp64]
};
