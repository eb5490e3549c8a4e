/*
 * gmmiv.h -- C ABI of libgmmiv, the MI355X (gfx950) GMM / i-vector compute engine.
 *
 * Drop-in boundary for ONE hot path of LIA_RAL: per-frame diagonal-GMM log-likelihood, top-C
 * selection, full-posterior (Baum-Welch / EM) sufficient statistics, the i-vector solve and the
 * i-vector scoring rules.  The reference has no FFI layer: these loops call C++ objects of the
 * external alize-core library once per frame.  Each entry point below therefore
 * replaces one *batched* reference loop; the comment on it cites the loop (paths relative to the
 * LIA_RAL tree).  INTEGRATION.md shows the branch a maintainer adds at each of those call sites.
 *
 * Conventions
 *  - plain C, opaque handles, int status (0 = ok, <0 = error; gmmiv_last_error() has the text);
 *  - every array argument may be a HOST pointer or a DEVICE (HIP) pointer -- detected with
 *    hipPointerGetAttributes.  Host arrays are staged through the context's device workspace;
 *    device arrays are used in place (no copy), which is what bench.py times;
 *  - matrices are row-major, arithmetic is fp64 like the reference (features may be given as
 *    float32, the on-disk SPro type, or as double, the type Feature::getDataVector() returns);
 *  - one context per GPU / per host thread; a context is not thread-safe, different contexts are;
 *  - no hidden state: the per-frame top-C vector that ALIZE keeps inside StatServer is an
 *    explicit output / input here.
 */
#ifndef GMMIV_H
#define GMMIV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmmiv_ctx gmmiv_ctx; /* device + stream + workspace                       */
typedef struct gmmiv_gmm gmmiv_gmm; /* device-resident diagonal GMM in kernel-ready layout */

enum { GMMIV_F32 = 0, GMMIV_F64 = 1 };           /* feature element type               */
enum { GMMIV_TOP_PARTIAL = 0, GMMIV_TOP_COMPLETE = 1 }; /* computeLLKWithTopDistribs */

#define GMMIV_OK 0
#define GMMIV_ERR_ARG (-1)
#define GMMIV_ERR_HIP (-2)
#define GMMIV_ERR_UNSUPPORTED (-3)
#define GMMIV_ERR_NUMERIC (-4)

/* ---- context --------------------------------------------------------------------------- */
/* stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL for a private non-blocking one.  A caller whose
 * other work runs on the NULL (legacy default) stream -- torch's default stream is that one -- passes GMMIV_STREAM_DEFAULT to
 * have the context launch there too (hipStreamLegacy, (hipStream_t)1, is taken to mean the same): a private stream is NOT
 * ordered with the NULL stream, so buffers written by NULL-stream work would have to be complete before each call. */
#define GMMIV_STREAM_DEFAULT ((void *)(intptr_t)-1)
int gmmiv_ctx_create(int device, void *stream, gmmiv_ctx **out);
void gmmiv_ctx_destroy(gmmiv_ctx *ctx);
int gmmiv_ctx_sync(gmmiv_ctx *ctx);
/* The hipStream_t every call of this context is enqueued on (the one given to gmmiv_ctx_create, or the private one): a host
 * layer that keeps its own device buffers orders its copies / memsets on it instead of synchronising around every call. */
void *gmmiv_ctx_stream(gmmiv_ctx *ctx);
const char *gmmiv_last_error(void);
const char *gmmiv_version(void);
/* Runtime knobs; returns the previous value (-1: unknown key).  EVERY option is state of the context it is set on -- two
 * contexts driven from one host thread keep their own settings, a context keeps its settings whichever thread drives it.
 *   "stats_z" 1        EM / Baum-Welch statistics from stored scaled likelihoods (k_llk_mfma<WZ> + k_stats_z);
 *                      0: the recomputing k_stats_mfma (also used for D > 60 or when the scratch does not fit)
 *   "z_scratch_mb"     likelihood scratch budget in MiB (default 16384, at most a quarter of the device's TOTAL memory):
 *                      frames are processed in chunks that fit.  The chunk length -- and with it the fp64 summation
 *                      order -- depends only on this option, the model shape and the device model, not on the memory
 *                      free at call time: results are bitwise reproducible across runs and ranks.  (The order differs
 *                      from the reference's frame-by-frame accumulation: parity is to a tolerance, see DESIGN.md.)
 *   "z_waves" 8        workgroup shape of k_stats_z (8, 16 or 4 waves)
 *   "z_depth_tv" 4     register sets of k_stats_z's likelihood stream (prefetch distance + 1; 2 or 4) in the N / F mode,
 *   "z_depth_em" 2     and in the EM mode; bit-identical results
 *   "prune_log2" 0     n > 0: skip groups of posteriors that are all below 2^-n (NOT the reference's arithmetic
 *                      for dead Gaussians; off by default)
 *   "tv_stats_split" 1 gmmiv_tv_stats on at most 16 utterances: every utterance in pieces of whole 64-frame tiles (more workgroups for the
 *                      N / F kernel), summed back in piece order; 0 = one segment per utterance.  The fp64 summation ORDER of an
 *                      utterance's row therefore depends on how many utterances share the call (pieces for <= 16, one segment above):
 *                      the same utterance extracted alone and inside a large batch gives N / F -- and i-vectors -- that agree to about
 *                      1e-13 relative, not bitwise (both within the 1e-9 of the parity tests); set the option to 0 when a row must not
 *                      depend on its neighbours
 *   "assume_finite" 0  1: skip the pass that COUNTS the frames with unusable feature values ("DEGENERATE INPUTS" below; results never depend on it)
 *   "screened_frames", "zero_llk_frames"   counters of the frames of kind (1) / kind (2) of "DEGENERATE INPUTS" (read: returns the
 *                      count so far and stores `value`)
 *   "tv_tett_direct" 1 estimateTETt as one kernel that computes the lower triangles only and writes them packed (D <= 64); 0 = batched
 *                      GEMM into full matrices + pack
 *   "tv_batch" 1024    utterances per batch of the i-vector solve / T-matrix E-step (one workgroup factors one
 *                      system L_u; workspace 4 x tv_batch x R^2 doubles)
 *   "tv_acc_mb" 8192   T-matrix E-step: MiB of packed E_u = L_u^-1 + w_u w_u^T kept in HBM before A += N^T E and Cmx += W^T F run
 *                      (one GEMM per super-batch, K = its utterances, instead of one per tv_batch)
 *   "chol_gemm" 0      1: the GEMM-built right-looking batched Cholesky / inverse instead of chol_fused.hip (always
 *                      used for odd orders); A/B switch (like "gemm_remap", "gemm_clamp", "gemm_narrow", "z_tv4")
 *   "gemm_nt80" 1      split-K NT products whose N is a multiple of 80 but not of 128 (aux = F (T Sigma^-1)^T at rank 400) on 128 x 80
 *                      tiles instead of 128 x 128 tiles + a 16-column strip; 0: the latter (A/B switch)
 *   "chol_lds" 1       chol_fused.hip stages the panel rows once per workgroup in LDS; 0: every wave fetches them itself (A/B switch)
 *   "chol_flow" 1      batched Cholesky k_chol_left2 (panel staged first, diagonal update from LDS on all waves); 0: round 2's k_chol_left
 *   "kopts_bound"      read-only: 1 when this context's kernel-launcher options are the set bound to the calling thread (they are
 *                      bound by each call of the context on entry)
 *   "tv_mstep_solve" 1 updateTestimate by blocked substitution through the Cholesky factor of A_c (k_chol_solve_multi);
 *                      0: explicit inverse + GEMM like the reference
 *   "tv_md_device" 1   minDivergence: R normalised and factored on the device (even R); 0: on the host
 *   "topc_fused" 1     DETERMINE_TOP_DISTRIBS with the candidates collected in the epilogue of the MFMA log-likelihood kernel
 *                      (k_llk_mfma<TC> + k_topc_rank; C' <= 16, C <= 2048, D <= 64); 0 or not applicable: "topc_z".
 *                      "topc_fallbacks" counts the calls the fused path handed on (candidate list overflow / margin check)
 *   "short_calls" 1    log-likelihood kernels: a call of at most 32 768 frames runs 4-wave workgroups (one round, one wave per SIMD: 0.3 ms
 *                      instead of 0.55 for the walk through a 2048-Gaussian model); 0 = the 8-wave workgroups of long calls.  Per-frame
 *                      results are the same either way.
 *   "topc_rank2" 1     fused path: the ranking kernel handles two frames per wave (k_topc_rank2; frames with more than 128 candidate
 *                      records or more than 32 survivors go through the one-frame kernel right behind it); 0 = one frame per wave
 *   "topc_use_lanes" 4  USE_TOP_DISTRIBS with at most 16 candidates: four lanes per candidate read 64 contiguous bytes of its model row per
 *                      instruction (k_topc_use4, one frame per wave); 1 = one lane per candidate (k_topc_use16, four frames per wave)
 *   "topc_rank_direct" 0  fused path: k_topc_rank ranks the survivors of the final threshold on their MFMA logits and re-evaluates them in
 *                      the reference's direct form only when another survivor lies within 1e-6 of a selected one (same selection and
 *                      order; selected likelihoods differ by < 1e-11 relative); 1 = direct form for every frame (the round-2 behaviour)
 *   "topc_z" 1         DETERMINE_TOP_DISTRIBS from the stored MFMA likelihoods (k_llk_mfma<WZ> + k_topc_from_z, direct form only
 *                      for the candidates); 0: the direct-form VALU kernel for every Gaussian
 *   "timing" 0         1: record HIP events around the kernels (gmmiv_ctx_kernel_ms)
 *   "glds", "wg_waves", "em_chunks", "dbg": A/B switches of the measurement tools */
long gmmiv_ctx_set_option(gmmiv_ctx *ctx, const char *key, long value);
/* Host callbacks at the two points of a TotalVariability iteration where an exchange can start before the call that produces its
 * payload has returned (the reference's threaded estimateAandC merges A, Cmx, R, r under one mutex AFTER all workers are done,
 * AccumulateTVStat.cpp:1920-1937; with one rank per GPU the 1.31 GB reduce-scatter of A can run under the Cmx GEMM instead):
 *   "tv_a_ready"    inside gmmiv_tv_estimate_a_and_c, right after the LAST `A += N^T E` GEMM has been enqueued on the context's
 *                   stream and before `Cmx += W^T F` is (device accumulators only) -- the callback typically calls
 *                   gmmiv_reduce_scatter_f64_begin on A;
 *   "md_factored"   inside gmmiv_tv_min_divergence, after R has been normalised and factored and before T is read -- the
 *                   callback joins an all-gather of T that was begun before the call (gmmiv_comm_join) and may finish T's layout.
 * The callback runs on the calling host thread; anything it enqueues on the context's stream is ordered like the library's own
 * work.  fn == NULL removes the hook.  Returns 0, or -1 for an unknown point. */
typedef void (*gmmiv_hook_fn)(void *user);
int gmmiv_ctx_set_hook(gmmiv_ctx *ctx, const char *point, gmmiv_hook_fn fn, void *user);
/* Duration (ms, HIP events on the context's stream) of the last call's dominant kernel. */
double gmmiv_ctx_last_kernel_ms(gmmiv_ctx *ctx, const char **kernel_name);
/* Same for a named kernel ("k_llk_mfma", "k_stats_z", "k_stats_mfma", ...): TOTAL over its launches
 * inside the most recent call that used it (a call may process its frames in several chunks);
 * -1 if none.  gmmiv_ctx_kernel_launches gives that number of launches. */
double gmmiv_ctx_kernel_ms(gmmiv_ctx *ctx, const char *kernel_name);
long gmmiv_ctx_kernel_launches(gmmiv_ctx *ctx, const char *kernel_name);

/* ---- model: MixtureGD / DistribGD ----------------------------------------------------------
 * w[C], mean[C*D], covinv[C*D] (DistribGD::getMeanVect / getCovInvVect, MixtureGD::weight(c);
 * LIA_SpkTools/src/AccumulateTVStat.cpp:154-162).  cst/det are derived like computeAll().
 * SHAPES.  mixtureDistribCount, vectSize and topDistribsCount are free configuration keys of the reference
 * (LIA_SpkDet/TrainWorld/cfg/TrainWorld.cfg, ComputeTest.cpp:129-215) and none of them is refused here: vectSize <= 80 runs the fp64
 * MFMA kernels (compiled for D <= 16 / 32 / 60 / 80; the stored-likelihood statistics path for D <= 60); a larger vectSize (<= 4096)
 * runs generic paths with the same results -- logits in the reference's direct form on the vector ALUs, statistics as
 * gamma^T [x | 1 | x^2] on the fp64 GEMM -- at about a tenth of the rate; gmmiv_tv_stats walks the UTTERANCES one by one there (per
 * utterance: posteriors, one C x (vectSize + 2) x length GEMM, a scatter -- five launches), so a call of many short utterances is
 * launch-bound on that path, not merely slower (vectSize 1 .. 80 never takes it).  The fused / stored-likelihood top-C selection serves
 * topDistribsCount <= 16 / <= 60, the LDS selection kernel <= 64 with up to ~4 700 Gaussians; anything beyond (8192 Gaussians,
 * topDistribsCount 100, ...) goes through an any-shape selection kernel whose logit rows live in device scratch. */
int gmmiv_gmm_create(gmmiv_ctx *ctx, int C, int D, const double *w, const double *mean,
                     const double *covinv, gmmiv_gmm **out);
int gmmiv_gmm_set(gmmiv_gmm *g, const double *w, const double *mean, const double *covinv);
/* Same from covariances (DistribGD::setCov + computeAll, TrainTools.cpp:577-582): covInv = 1/cov. */
int gmmiv_gmm_set_cov(gmmiv_gmm *g, const double *w, const double *mean, const double *cov);
void gmmiv_gmm_destroy(gmmiv_gmm *g);

/* ---- DEGENERATE INPUTS: what every frame-consuming entry point does with them ------------------------------------------
 * The reference has no defined behaviour here (a NaN feature poisons every accumulator it touches; ALIZE's handling of a frame of
 * likelihood 0 is not visible from LIA_RAL -- SURVEY.md U1; the one in-tree guard, TopGauss.cpp:247, maps a NaN likelihood to
 * exp(minLLK)).  This library defines it, and tests/test_gpu_degenerate.py holds every path to it:
 *
 * A frame is a ZERO-LIKELIHOOD FRAME when (1) one of its feature values is NaN, infinite or larger than 1e18 in magnitude, or
 * (2) its likelihood is 0 in fp64: the LARGEST term w_c lk_c(x_t) lies below 2^-1075 = exp(-745.13) (so every term of the
 * reference's linear-domain sum rounds to 0 and log of it is -inf), or the log-sum is not finite.  The decision is made on the
 * largest logit, threshold log 2^-1075 = -745.1332 (GMMIV_ZERO_LLK in csrc/devutil.h), on every path; a frame whose best Gaussian
 * is at -744 is an ordinary frame, one at -746 is a zero-likelihood frame (tests/test_gpu_degenerate.py pins both sides).  Then
 *   gmmiv_llk                     llk_t = min_llk (the clamp of log 0); counted in sums like any frame
 *   gmmiv_llk_determine_top       idx = 0 .. ctop-1 (the tie rule -- lowest index first -- on equal, zero, likelihoods), lk = 0,
 *                                 nontop_lk = 0, nontop_llk = -inf, nontop_w = 1 - sum of those weights, llk = min_llk
 *   gmmiv_topgauss_compute        as determine_top; count = cap for a mass threshold (the reference's loop runs to the end), idx as above
 *   gmmiv_llk_use_top(_multi)     llk_t = min_llk
 *   gmmiv_occ                     a row of zeros
 *   gmmiv_em_accumulate           the frame adds NOTHING: no occupancy, no first / second order statistics, nothing to the sum of
 *                                 log-likelihoods and nothing to the frame count (the M-step weights still sum to 1)
 *   gmmiv_tv_stats(_lines), gmmiv_jfa statistics    nothing added to N / F
 *   gmmiv_frame_moments           NOT screened: sums of the raw values, a NaN goes into the sums like in the reference
 * Frames of kind (1) are handled ON THE DEVICE, inside the kernels: every kernel that reads features reads an unusable value as 1e10
 * (csrc/devutil.h, feat_sane: one compare + select where the value is loaded -- in the MFMA log-likelihood kernel once per frame and
 * workgroup, outside its loop).  The value is finite, so no 0 x NaN reaches a statistic, and it puts every logit of the frame near
 * -0.5e20 / variance: the frame then IS a frame of kind (2) for every kernel (holds for variances in 1e-17 .. 1e17) and follows the
 * table above with no host decision -- no flags read back, no compaction, NO SYNCHRONISATION: a call whose arrays are all device
 * pointers only enqueues on the context's stream (rounds 1-5 screened on the host and waited for the stream once per call).
 * What remains of the screening is a COUNT: at the start of a frame-consuming call one pass over x (0.3 % of an EM pass) adds the number
 * of kind-(1) frames to a device counter, option "screened_frames" (read / reset like "zero_llk_frames" below).  The option
 * "assume_finite" 1 skips that pass; RESULTS do not depend on it (the C++ host layer checks a FeatureBuffer once, at upload, and
 * sets it per call from the buffer the call reads).  Kind (2) is decided per frame where the log-likelihood kernel finishes a
 * frame -- no per-element work in the hot loops -- and COUNTED on the device by the entry points that drop such a frame from a sum:
 * gmmiv_llk, gmmiv_em_accumulate, gmmiv_tv_stats(_lines) (and the JFA statistics built on it), gmmiv_occ; a kind-(1) frame is
 * evaluated as a kind-(2) frame and therefore counted here TOO.  gmmiv_ctx_set_option(ctx, "zero_llk_frames", v) returns the count so
 * far and stores v (0 to reset); the read waits for the context's stream, the counting never does.  Not counted: the top-C entry
 * points (a zero-likelihood frame is visible there as llk = min_llk with lk = 0).
 * Other edges: T = 0 is valid everywhere (outputs untouched, accumulators unchanged); a Gaussian of weight 0 has likelihood 0
 * (never selected before a Gaussian of positive likelihood, occupancy 0); gmmiv_em_get keeps the previous mean / covariance of a
 * Gaussian whose occupancy is 0 and gives it weight 0; identical Gaussians tie and the lower index wins.
 *
 * gmmiv_count_unusable_frames: the counting pass on its own, read back -- *count = frames of kind (1) (this call waits for the stream). */
int gmmiv_count_unusable_frames(gmmiv_ctx *ctx, const void *x, int x_dtype, int64_t T, int64_t ldx, int D,
                                int64_t *count);

/* ---- FrameAccGD::accumulate loop (LIA_SpkTools/src/AccumulateStat.cpp:387-396) ---------------
 * acc[0..D) += sum x, acc[D..2D) += sum x^2, acc[2D] += T.  mean/cov: sum/n, sumsq/n - mean^2. */
int gmmiv_frame_moments(gmmiv_ctx *ctx, const void *x, int x_dtype, int64_t T, int64_t ldx, int D,
                        double *acc);

/* ---- frame selection on the device: out[i][0..D) = x[frame_idx[i]][0..D) ----------------------
 * Replaces the seekFeature/readFeature walk over a SegCluster (label selection, bagging:
 * LIA_SpkTools/src/AccumulateStat.cpp:121-128, GeneralTools.cpp:455-510) for features that stay
 * resident in HBM.  x, out: DEVICE arrays (out has ld = D); frame_idx: host or device. */
int gmmiv_gather_frames(gmmiv_ctx *ctx, const void *x, int x_dtype, int64_t ldx, int D,
                        const int64_t *frame_idx, int64_t n, void *out);
/* The same selection given as RUNS of adjacent frames -- what a SegCluster is (a bagged chunk of
 * baggedSegments is 3..7 frames, GeneralTools.cpp:455-510; a label segment thousands):
 * runs[3 r + 0 .. 2] = (first source frame, first output row, length); run r copies frames
 * [src, src + len) to rows [dst, dst + len) of out.  24 bytes per run cross PCIe instead of 8 per
 * frame.  Runs must not overlap in `out`; long runs should be cut into pieces of <= 64 frames by
 * the caller (one wavefront moves one run).  x, out: DEVICE arrays; runs: host or device -- with a
 * device table the call only enqueues on the context's stream (no synchronisation). */
int gmmiv_gather_runs(gmmiv_ctx *ctx, const void *x, int x_dtype, int64_t ldx, int D,
                      const int64_t *runs, int64_t nrun, void *out);

/* ---- segment means of per-frame values (ComputeTest's score per segment, ComputeTest.cpp:181-199) ------
 * out[r * nseg + s] = mean of v[r * ld + t], t in [seg_begin[s], seg_begin[s+1])  (0 for an empty segment).
 * v: DEVICE array of nrows rows (the world's and the clients' per-frame log-likelihoods as the llk_* entry points
 * leave them); seg_begin: nseg + 1 HOST offsets; out: host or device.  The summation order is fixed by the
 * segment bounds alone (pieces of 8192 frames, added in order). */
int gmmiv_segment_means(gmmiv_ctx *ctx, const double *v, int64_t ld, int nrows, const int64_t *seg_begin,
                        int64_t nseg, double *out);

/* ---- MixtureStat::computeAndAccumulateLLK(f,1.0,TOP_DISTRIBS_NO_ACTION) loop -------------------
 * (LIA_SpkTools/src/AccumulateStat.cpp:69-94, :344-379; AccumulateTVStat.cpp:1644-1648)
 * llk_out[T] (nullable) = clamp(log sum_c w_c lk_c(x_t), min_llk, max_llk);
 * sums[0] += sum_t llk_out[t] (clamped), sums[1] += T  -> getMeanLLK() = sums[0]/sums[1]. */
int gmmiv_llk(gmmiv_ctx *ctx, const gmmiv_gmm *g, const void *x, int x_dtype, int64_t T, int64_t ldx,
              double min_llk, double max_llk, double *llk_out, double *sums);

/* ---- computeAndAccumulateLLK(f,1.0,DETERMINE_TOP_DISTRIBS) + StatServer::getTopDistribIndexVector
 * (LIA_SpkDet/ComputeTest/src/ComputeTest.cpp:163; LIA_SpkTools/src/TopGauss.cpp:167-193)
 * Per frame: idx[t*ctop+j] / lk[t*ctop+j] = the ctop largest w_c lk_c, descending (ties: lower
 * index first); nontop_lk = sum of the others (linear, may underflow), nontop_llk = its log
 * (-inf when empty), nontop_w = 1 - sum of the selected weights; llk = clamp(log(top [+ rest])).
 * lk, nontop_lk, nontop_w, llk_out may be NULL.  1 <= ctop <= min(64, C): a request for more Gaussians than the
 * model has is GMMIV_ERR_ARG, not clamped -- the row stride of idx / lk is the caller's ctop, so the caller clamps. */
int gmmiv_llk_determine_top(gmmiv_ctx *ctx, const gmmiv_gmm *world, const void *x, int x_dtype,
                            int64_t T, int64_t ldx, int ctop, int mode, double min_llk, double max_llk,
                            int32_t *idx, double *lk, double *nontop_lk, double *nontop_llk,
                            double *nontop_w, double *llk_out);

/* ---- TopGauss::compute (LIA_SpkTools/src/TopGauss.cpp:136-198): the per-frame Gaussian selection that the factor-analysis
 * tools store per feature file.  The sorted top list of every frame comes from DETERMINE_TOP_DISTRIBS with topDistribsCount =
 * cap (1 <= cap <= min(64, C); the reference's "this should be high enough"), then
 *   top_gauss >= 1 : count[t] = (int)top_gauss                                            (:170)
 *   top_gauss <  1 : Gaussians are taken, heaviest first, until their cumulative likelihood exceeds top_gauss * exp(llk_t)
 *                    (the test precedes each addition, :163-167) -- a VARIABLE count per frame, at most cap; *n_capped
 *                    (HOST, nullable) = frames whose mass was not reached within cap entries;
 *   snsw[t] = 1 - sum of the selected weights, snsl[t] = max(exp(llk_t) - sum of the selected likelihoods, EPS_LK = 1e-200).
 * idx [T x cap]: the selected indices of frame t first, -1 behind them -- gmmiv_llk_use_top(ctop = cap, idx, log(snsl)) then
 * evaluates exactly the stored selection (TopGauss::get, :275-316).  llk_out (nullable) = the clamped per-frame llk of the
 * DETERMINE pass (its mean is what compute() returns).  All arrays host or device. */
int gmmiv_topgauss_compute(gmmiv_ctx *ctx, const gmmiv_gmm *ubm, const void *x, int x_dtype, int64_t T, int64_t ldx,
                           int cap, double top_gauss, int mode, double min_llk, double max_llk, int32_t *idx,
                           int32_t *count, double *snsw, double *snsl, double *llk_out, int64_t *n_capped);

/* ---- computeAndAccumulateLLK(f,1.0,USE_TOP_DISTRIBS) on a client model ------------------------
 * (ComputeTest.cpp:166-167; StatServer::setTopDistribIndexVector, TopGauss.cpp:297-308)
 * llk_out[t] = clamp(log(sum_j w_c lk_c(client) over c = idx[t][j]  [+ exp(nontop_llk[t])])).
 * ctop as above; an entry of idx outside [0, C) contributes nothing (it is never dereferenced). */
int gmmiv_llk_use_top(gmmiv_ctx *ctx, const gmmiv_gmm *client, const void *x, int x_dtype, int64_t T,
                      int64_t ldx, int ctop, const int32_t *idx, const double *nontop_llk, int mode,
                      double min_llk, double max_llk, double *llk_out);

/* The same for SEVERAL client models on one test segment -- the client loop of ComputeTest.cpp:170-207 (every model of an ndx line
 * is scored on the same frames with the same world indices) as ONE call: llk_out is [n_clients][T] (host or device), row i what
 * gmmiv_llk_use_top returns for clients[i].  All clients must have the dimension count of clients[0]; one launch when the
 * four-lanes-per-candidate kernel applies (ctop <= 16, even dimension count), client by client otherwise. */
int gmmiv_llk_use_top_multi(gmmiv_ctx *ctx, int n_clients, const gmmiv_gmm *const *clients, const void *x, int x_dtype, int64_t T,
                            int64_t ldx, int ctop, const int32_t *idx, const double *nontop_llk, int mode,
                            double min_llk, double max_llk, double *llk_out);

/* ---- MixtureGDStat::computeAndAccumulateOcc + getOccVect (AccumulateTVStat.cpp:302,334-335;
 * FactorAnalysis.cpp:204-205): the full posterior vector of every frame,
 * gamma[t*C + c] = w_c lk_c(x_t) / sum_c' w_c' lk_c'(x_t)   (row-major [T x C]). */
int gmmiv_occ(gmmiv_ctx *ctx, const gmmiv_gmm *gmm, const void *x, int x_dtype, int64_t T, int64_t ldx,
              double *gamma);

/* ---- MixtureStat::computeAndAccumulateEM loop (accumulateStatEM, AccumulateStat.cpp:103-152) ---
 * acc is the flat EM accumulator, length gmmiv_em_acc_len(C,D) = C*(1+2D)+2 doubles:
 *   [ occ[C] | sum g x [C*D] | sum g x^2 [C*D] | sum_t weight*log lk_t | sum_t weight ].
 * ACCUMULATES (resetEM = zero the array; addAccEM = add two arrays; one RCCL all-reduce of this
 * array merges ranks).  gamma_tc = full posterior (all C). */
size_t gmmiv_em_acc_len(int C, int D);
int gmmiv_em_accumulate(gmmiv_ctx *ctx, const gmmiv_gmm *g, const void *x, int x_dtype, int64_t T,
                        int64_t ldx, double weight, double *acc);
/* MixtureStat::getEM(): w = occ/count, mean = sx/occ, cov = sxx/occ - mean^2 (host formula run on
 * the device copy; components with occ == 0 keep prev_mean / prev_cov).  Outputs [C],[C*D],[C*D]. */
int gmmiv_em_get(gmmiv_ctx *ctx, int C, int D, const double *acc, const double *prev_mean,
                 const double *prev_cov, double *w, double *mean, double *cov);

/* varianceControl (LIA_SpkTools/src/TrainTools.cpp:567-587): cov[c,d] clamped to
 * [flooring*cov_signal[d], ceiling*cov_signal[d]] in place (floor test first, then ceiling);
 * counts (HOST, nullable): [0] += floored entries, [1] += ceiled entries. */
int gmmiv_variance_control(gmmiv_ctx *ctx, int C, int D, double *cov, double flooring, double ceiling,
                           const double *cov_signal, int64_t *counts);

/* ---- TVAcc::computeAndAccumulateTVStat (LIA_SpkTools/src/AccumulateTVStat.cpp:281-351) ---------
 * Frames of statistics row u are x[utt_begin[u] .. utt_begin[u+1]) (utt_begin: U+1 HOST offsets).
 * N[u*C+c] = sum_t g_tc ; F[u*C*D + c*D + i] = sum_t g_tc x_ti   (rows are overwritten). */
int gmmiv_tv_stats(gmmiv_ctx *ctx, const gmmiv_gmm *g, const void *x, int x_dtype, int64_t T,
                   int64_t ldx, const int64_t *utt_begin, int64_t U, double *N, double *F);
/* The same with the reference's file -> ndx-line map (AccumulateTVStat.cpp:318-346, TVTranslate::locIndices): the frames of
 * feature file f are x[file_begin[f] .. file_begin[f+1]); statistics row l (an ndx line) is the SUM over the files
 * line_files[line_off[l] .. line_off[l+1]) -- a file listed on several lines is evaluated once and added to each of them
 * (file_begin, line_off, line_files: HOST arrays; rows are overwritten). */
int gmmiv_tv_stats_lines(gmmiv_ctx *ctx, const gmmiv_gmm *g, const void *x, int x_dtype, int64_t T, int64_t ldx,
                         const int64_t *file_begin, int64_t nfiles, int64_t nlines, const int64_t *line_off,
                         const int64_t *line_files, double *N, double *F);

/* ---- TVAcc i-vector maths (exact mode) -------------------------------------------------------
 * T: [R x C*D] row-major total-variability matrix; invvar: [C*D] UBM inverse variances.
 * substractM          AccumulateTVStat.cpp:1088-1105   F[u,c,:] -= mean[c,:] N[u,c]  (in place)
 * estimateTETt        :777-805    TETt packed lower triangle: [C x R(R+1)/2], row i>=j at i(i+1)/2+j
 * estimateW           :2114-2169  W[U x R] = (I + sum_c N[u,c] TETt_c)^-1 T Sigma^-1 F_u
 * estimateAandC       :1702-1795  + A[C x R(R+1)/2] (packed), Cmx[R x C*D], Rm[R x R], r[R], meanW[R]
 *                     (A, Cmx, Rm, r, meanW ACCUMULATE: zero them first; meanW is the SUM of w --
 *                      divide by the total utterance count after the all-reduce)
 * updateTestimate     :974-1005   T_c = A_c^-1 Cmx_c
 * minDivergence       :2056-2099  T <- chol_upper(Rm/n - r r^T / n^2) T ; mean += T^T meanW
 */
int gmmiv_tv_subtract_m(gmmiv_ctx *ctx, int64_t U, int C, int D, const double *N, double *F,
                        const double *ubm_means);
/* F_dst = F_src - N ubm_means, out of place (F_dst may be F_src): TotalVariability reloads N / F and calls substractM at the top of
 * every iteration (TotalVariability.cpp:123-124, substractM works in place); with the pristine statistics kept in HBM the restore
 * and the centring are ONE pass over F instead of a copy and a read-modify-write. */
int gmmiv_tv_subtract_m_to(gmmiv_ctx *ctx, int64_t U, int C, int D, const double *N, const double *F_src, double *F_dst,
                           const double *ubm_means);
size_t gmmiv_tv_packed_len(int R);
int gmmiv_tv_tett(gmmiv_ctx *ctx, int C, int D, int R, const double *Tm, const double *invvar,
                  double *tett_packed);
int gmmiv_tv_estimate_w(gmmiv_ctx *ctx, int64_t U, int C, int D, int R, const double *N,
                        const double *F, const double *Tm, const double *invvar,
                        const double *tett_packed, double *W);
int gmmiv_tv_estimate_a_and_c(gmmiv_ctx *ctx, int64_t U, int C, int D, int R, const double *N,
                              const double *F, const double *Tm, const double *invvar,
                              const double *tett_packed, double *W, double *A_packed, double *Cmx,
                              double *Rm, double *r, double *meanW);
int gmmiv_tv_update_t(gmmiv_ctx *ctx, int C, int D, int R, const double *A_packed, const double *Cmx,
                      double *Tm);
int gmmiv_tv_min_divergence(gmmiv_ctx *ctx, int C, int D, int R, double n_sessions, double *Rm,
                            double *r, const double *meanW, double *ubm_means, double *Tm);
/* orthonormalizeT (AccumulateTVStat.cpp:1548-1596): classical Gram-Schmidt over the rows of
 * T[R x SV], in place (coefficients taken against the ORIGINAL row, zero rows stay zero). */
int gmmiv_tv_orthonormalize_t(gmmiv_ctx *ctx, int R, int64_t SV, double *Tm);

/* ---- approximate extractors (IvExtractor modes ubmWeight / eigenDecomposition, IvExtractor.cpp:150-360) ----
 * normStatistics (AccumulateTVStat.cpp:1225-1242): F[u,c,d] = (F - mean[c,d] N[u,c]) sqrt(invvar[c,d]), in place.
 * substractMplusTW (:1379-1399, getMplusTW :964-971): F[u,c,d] -= (mean[c,d] + sum_i T[i,cD+d] W[u,i]) N[u,c].
 * normTMatrix (:1600-1609): T[j,k] *= sqrt(invvar[k]), in place.
 * getWeightedCov (:2837-2855): Wm[R x R] = sum_c weight[c] T_c T_c^T.
 * approximateTcTc (:3116-3136): Dm[c,i] += || (T_c^T Q)[:, i] ||^2, Dm [C x R], Q [R x R]; accumulates like the
 *   reference (zero Dm first for a fresh result).  Q comes from computeEigenProblem (:2997-3102, Eigen / LAPACK
 *   on the host in the reference; its column order is the solver's, so Q is an input here).
 * estimateWUbmWeight (:2348-2396): W[u] += (I + (sum_c N[u,c]) Wm)^-1 (T F[u]); T, F normalised as above.
 * estimateWEigenDecomposition (:2566-2609): W[u] += Q diag(1 / (1 + N[u] Dm)) Q^T (T F[u]).
 * W is accumulated into (the reference zeroes _W in ubmWeight mode only: pass zeros for a fresh result). */
int gmmiv_tv_norm_statistics(gmmiv_ctx *ctx, int64_t U, int C, int D, const double *N, double *F,
                             const double *ubm_means, const double *invvar);
int gmmiv_tv_subtract_m_plus_tw(gmmiv_ctx *ctx, int64_t U, int C, int D, int R, const double *N, double *F,
                                const double *ubm_means, const double *Tm, const double *W);
int gmmiv_tv_norm_t(gmmiv_ctx *ctx, int C, int D, int R, double *Tm, const double *invvar);
int gmmiv_tv_weighted_cov(gmmiv_ctx *ctx, int C, int D, int R, const double *Tm, const double *weight, double *Wm);
int gmmiv_tv_approximate_tctc(gmmiv_ctx *ctx, int C, int D, int R, const double *Tm, const double *Q, double *Dm);
int gmmiv_tv_estimate_w_ubm_weight(gmmiv_ctx *ctx, int64_t U, int C, int D, int R, const double *N, const double *F,
                                   const double *Tm, const double *Wm, double *W);
int gmmiv_tv_estimate_w_eigen(gmmiv_ctx *ctx, int64_t U, int C, int D, int R, const double *N, const double *F,
                              const double *Tm, const double *Dm, const double *Q, double *W);

/* ---- PldaTest::center / rotateLeft / lengthNorm (PldaTools.cpp:3706-3790); one iteration of
 * sphericalNuisanceNormalization (:3793-3839) = all three ------------------------------------------
 * Y[dim_out x n] = lengthNorm( M[dim_out x dim_in] * (X[dim_in x n] - mean[dim_in]) ), vectors as
 * columns.  mean == NULL: no centring; M == NULL: no rotation (dim_out == dim_in);
 * length_norm == 0: columns are not normalised.  Y may alias X when M == NULL. */
int gmmiv_iv_normalize(gmmiv_ctx *ctx, int dim_in, int dim_out, int64_t n, const double *X,
                       const double *mean, const double *M, int length_norm, double *Y);

/* ---- i-vector back-end estimation on a development set: PldaDev (LIA_SpkTools/src/PldaTools.cpp) ----------------
 * X [dim x n], one vector per column like PldaDev::_data; sessions are grouped by speaker and
 * sessions_per_speaker[nspk] (host array, PldaDev::_session_per_speaker) gives the group sizes (sum = n, all > 0).
 *   gmmiv_dev_means        computeAll                 :353-387   mean[dim], spk_means[dim x nspk]
 *   gmmiv_dev_cov_mat      computeCovMat              :527-566   Sigma, W, B [dim x dim], all divided by n
 *   gmmiv_dev_wccn_chol    computeWccnChol            :1124-1176 upperCholesky((mean_c cov_c / n_c)^-1)
 *   gmmiv_dev_mahalanobis  computeMahalanobis         :1366-1378 W^-1
 *   gmmiv_dev_scatter_mat  computeScatterMat          :1610-1644 as written in the reference (SB unweighted and
 *                          unnormalised; SW = matrix of the LAST speaker, built from the first n_c sessions of the set)
 *   gmmiv_sym_eigen        computeEigenProblem        :1490-1535 for a SYMMETRIC matrix: vect[n x rank] (columns =
 *                          eigenvectors), val[rank] descending (host Jacobi; the reference calls Eigen / LAPACK)
 *   gmmiv_dev_efr_matrix   sphericalNuisanceNormalization :1852-1902: (V diag(lambda^-1/2))^T of Sigma (EFR) or W (sphNorm)
 *   gmmiv_dev_lda          computeLDA                 :1381-1413 rank leading eigenvectors (unit norm) of W^-1 B as rows
 * Any null output pointer is skipped.  The O(dim^3) pieces (eigen problems, the WCCN Cholesky) run on the host like
 * the reference's; the O(dim^2 n) covariance GEMMs run on the device. */
int gmmiv_dev_means(gmmiv_ctx *ctx, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                    double *mean, double *spk_means);
int gmmiv_dev_cov_mat(gmmiv_ctx *ctx, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                      double *Sigma, double *W, double *B);
int gmmiv_dev_wccn_chol(gmmiv_ctx *ctx, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                        double *WCCN);
int gmmiv_dev_mahalanobis(gmmiv_ctx *ctx, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                          double *M);
int gmmiv_dev_scatter_mat(gmmiv_ctx *ctx, int dim, int64_t n, const double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                          double *SB, double *SW);
int gmmiv_sym_eigen(gmmiv_ctx *ctx, int n, const double *A, int rank, double *vect, double *val);
int gmmiv_dev_efr_matrix(gmmiv_ctx *ctx, int dim, const double *Cov, double *M);
int gmmiv_dev_lda(gmmiv_ctx *ctx, int dim, const double *W, const double *B, int rank, double *ldaMat, double *eigval);

/* PldaModel::em_iteration (PldaTools.cpp:2329-2343 = center(Delta) + computeCovMatEigen :931-950 +
 * getExpectedValues :2359-2484 + mStep :2790-2815): one EM iteration of the PLDA model
 *   x = mu + F h_spk + G w_session + eps,  eps ~ N(0, Sigma)
 * on the development set X [dim x n] (sessions grouped by speaker; X is centred IN PLACE by the incoming Delta, like
 * _Dev.center(_Delta)).  F [dim x rf], G [dim x rg], Sigma [dim x dim], Delta [dim] are updated in place (M-step with
 * the minimum-divergence re-scaling of F and G).  rg may be 0 (pldaEigenChannelNumber 0, the simplified model: G is then
 * ignored and may be NULL).  The O(dim n r) products run on the device, the per-speaker r x r algebra on the host like the
 * reference's Eigen code. */
int gmmiv_plda_em_iteration(gmmiv_ctx *ctx, int dim, int64_t n, double *X, int64_t nspk, const int64_t *sessions_per_speaker,
                            int rf, int rg, double *F, double *G, double *Sigma, double *Delta);

/* PldaModel::preComputation + FTJ / FTJF of pldaNativeScoring (PldaTools.cpp:2950-2972, 4494-4496):
 *   FTJ[rf x dim] = F^T S^-1 - F^T S^-1 G (G^T S^-1 G + I)^-1 G^T S^-1,  FTJF[rf x rf] = FTJ F
 * F [dim x rf], G [dim x rg] (rg may be 0), Sigma [dim x dim] symmetric positive definite, all row-major.
 * rotateLeft(FTJ) is gmmiv_iv_normalize with FTJ as the matrix; FTJF feeds gmmiv_score_plda. */
int gmmiv_plda_precompute(gmmiv_ctx *ctx, int dim, int rf, int rg, const double *F, const double *G,
                          const double *Sigma, double *FTJ, double *FTJF);
/* The two-covariance model of PldaTest::twoCovScoring (PldaTools.cpp:4089-4125):
 *   G = W^-1 (B^-1 + 2 W^-1)^-1 W^-1,   H = W^-1 (B^-1 + W^-1)^-1 W^-1     (W, B symmetric positive definite) */
int gmmiv_twocov_model(gmmiv_ctx *ctx, int dim, const double *W, const double *B, double *G, double *H);
/* ---- PldaTest scoring (LIA_SpkTools/src/PldaTools.cpp) ------------------------------------------
 * models[dim x M], segs[dim x S]: one vector per COLUMN like PldaTest::_models/_segments;
 * scores[M x S].
 * cosineDistance :3842-3879, mahalanobisDistance :3882-3909 (-1/2 (m-s)^T Mah (m-s)),
 * twoCovScoring :4127-4171 ((m+s)^T G (m+s) - m^T H m - s^T H s),
 * pldaScoringUnThreaded :4186-4271 on vectors already projected by FTJ; models = per-speaker sums,
 * nsess[m] = enrolment sessions of model m (HOST array). */
int gmmiv_score_cosine(gmmiv_ctx *ctx, int dim, int64_t M, int64_t S, const double *models,
                       const double *segs, double *scores);
int gmmiv_score_mahalanobis(gmmiv_ctx *ctx, int dim, int64_t M, int64_t S, const double *models,
                            const double *segs, const double *Mah, double *scores);
int gmmiv_score_twocov(gmmiv_ctx *ctx, int dim, int64_t M, int64_t S, const double *models,
                       const double *segs, const double *G, const double *H, double *scores);
int gmmiv_score_plda(gmmiv_ctx *ctx, int rf, int64_t M, int64_t S, const double *models_sum,
                     const int64_t *nsess, const double *segs, const double *FTJF, double *scores);
/* PldaTest::twoCovScoringMixPart (:3923-3949, the L3 seam of twoCovScoring): scores[m][s] += (m + s)^T G (m + s) for every
 * pair -- ACCUMULATES like the reference's `_scores(m,s) +=` (zero the array for the bare term). */
int gmmiv_score_twocov_mix_part(gmmiv_ctx *ctx, int dim, int64_t M, int64_t S, const double *models, const double *segs,
                                const double *G, double *scores);
/* PldaTest::_trials (:3437, :3591-3620): cosineDistance / mahalanobisDistance score only the listed trials (:3871, :3889), the
 * other cells keep _scores' initial 0.  All rules above fill the whole M x S block with one GEMM; this call then writes `fill`
 * into every cell whose flag trials[m*S + s] is 0 (bytes, host or device). */
int gmmiv_score_apply_trials(gmmiv_ctx *ctx, int64_t M, int64_t S, const unsigned char *trials, double fill, double *scores);
/* Multi-GPU scoring: the M x S matrix tiles by blocks of MODELS, no collective -- rank g calls any rule above
 * with the columns [m0, m1) of `models` (and the matching nsess / rows of scores); gmmiv_shard_range gives the contiguous range
 * of rank `rank` out of `world` over n items (sizes differ by at most one), the same split as the reference's thread ranges
 * (PldaTools.cpp:4175-4183 dispatch, AccumulateTVStat.cpp:498-507). */
void gmmiv_shard_range(int64_t n, int rank, int world, int64_t *begin, int64_t *end);

/* ---- JFA (LIA_SpkTools/src/AccumulateJFAStat.cpp): model M_{s,h} = m + V y_s + U x_h + D z_s ---------------------------
 * The factor steps are the total-variability entry points under the JFA names:
 *   JFAAcc::estimateVEVT / estimateUEUT (:1266-1352, :1425-1508)                         -> gmmiv_tv_tett
 *   JFAAcc::estimateAndInverseL_EV + estimateYandV (:1970-1996, :2467-2511), _EC + estimateXandU (:2137-2163, :3040-3083)
 *                                                                                          -> gmmiv_tv_estimate_a_and_c
 *   JFAAcc::estimateY / estimateX (:2867-2957, :3262-3351)                                -> gmmiv_tv_estimate_w
 *   JFAAcc::updateVestimate / updateUestimate (:3597-3644)                                -> gmmiv_tv_update_t
 * What is specific to JFA is below.  All arrays host or device. */
/* F[r,c,:] -= N[r,c] (means[c,:] + (W[o] T)[c,:] + Dm[c,:] Z[o][c,:]),  o = owner ? owner[r] : r.  Every term is optional
 * (means, T/W, Dm/Z may be NULL).  N [rows x C], F [rows x C*D], T [R x C*D], W [nfact x R], Dm [C*D], Z [nfact x C*D].
 * Replaces JFAAcc::substractMplusDZ (:3805-3822), substractMplusVY (:3988-4005), substractMplusVYplusDZ (:4400-4422, owner =
 * the speaker of each session), the mean part of substractMplusUX (:4336-4364; its channel part is gmmiv_jfa_subtract_sessions),
 * getMplusVYplusDZ / getUX (:1803-1957). */
int gmmiv_jfa_subtract(gmmiv_ctx *ctx, int64_t rows, int C, int D, const double *N, double *F, const int64_t *owner, int64_t nfact,
                       const double *means, int R, const double *T, const double *W, const double *Dm, const double *Z);
/* F_X[s,c,:] -= sum over the sessions h in [sess_begin[s], sess_begin[s+1]) of N_h[h,c] (x_h U)[c,:]   (sessions grouped by
 * speaker, sess_begin a HOST array of nspk + 1 offsets).  Replaces JFAAcc::substractUX (:4152-4172). */
int gmmiv_jfa_subtract_sessions(gmmiv_ctx *ctx, int64_t nspk, const int64_t *sess_begin, int C, int D, const double *N_h, double *F_X,
                                int R, const double *U, const double *X);
/* tau < 0: Z = F iv D / (1 + N iv D^2) (JFAAcc::estimateZ, :3550-3573); tau >= 0: Z = tau / (tau + N) D iv F (estimateZMAP, :3576-3594). */
int gmmiv_jfa_estimate_z(gmmiv_ctx *ctx, int64_t nspk, int C, int D, const double *N, const double *F, const double *invvar, const double *Dm,
                         double tau, double *Z);
/* JFAAcc::estimateZandD (:3480-3516): Z as above (tau < 0) and D <- sum_s z F / sum_s (1 / L + z^2) N, in place. */
int gmmiv_jfa_estimate_z_and_d(gmmiv_ctx *ctx, int64_t nspk, int C, int D, const double *N, const double *F, const double *invvar, double *Dm,
                               double *Z);

/* ---- collectives of the paths that shard: RCCL over xGMI ---------------------------------------------
 * The reference merges the private accumulators of its worker threads under a mutex: MixtureStat::addAccEM
 * (LIA_SpkTools/src/AccumulateStat.cpp:286-292) for the EM statistics, `+=` of A / Cmx / R / r in the threaded
 * estimateAandC (AccumulateTVStat.cpp:1920-1937, 2036-2044).  With one rank per GPU the merge is a collective on the
 * accumulators where they lie (device memory):
 *   EM statistics (TrainWorld)        gmmiv_allreduce_f64 of the flat accumulator, gmmiv_em_acc_len() doubles (1.98 MB)
 *   T-matrix EM (TotalVariability)    gmmiv_reduce_scatter_f64 of A_packed and Cmx by blocks of Gaussians -> each rank
 *                                     solves T_c = A_c^-1 Cmx_c for its own Gaussians (updateTestimate is independent per
 *                                     Gaussian, :981-1000) -> gmmiv_allgather_f64 of the T blocks; R, r, meanW: allreduce
 * One communicator per context.  Rank 0 obtains an id (gmmiv_comm_get_unique_id) and ships its 128 bytes to the other ranks
 * over any host channel (gmmiv_comm_exchange_id_file: a file in a directory all ranks see; MPI / a TCP store work as well),
 * then EVERY rank calls gmmiv_comm_create (collective).  world == 1 needs no id and no RCCL.  The calls are enqueued on the
 * context's stream; device buffers are used in place, host buffers (allreduce / broadcast only) are staged and the call
 * returns after the result is back.  RCCL is loaded at run time (dlopen of the copy already mapped in the process, else
 * librccl.so.1; the environment variable GMMIV_RCCL_LIB, when set, names the ONLY library tried): GMMIV_ERR_UNSUPPORTED when
 * it cannot be found.
 *
 * Transports.  The id rank 0 draws selects the transport of the communicator every rank then creates from it:
 *   "rccl"  (default) RCCL over xGMI / PCIe, one rank per GPU -- the production path;
 *   "shm"   ranks of ONE host that may SHARE a GPU (RCCL refuses two ranks on one device), or a host without RCCL: buffers
 *           are staged through one mmap'ed file (GMMIV_COMM_SHM_DIR, default /dev/shm; GMMIV_COMM_SHM_SLOT_MB per rank,
 *           default 16) and every rank sums the pieces on its own device in rank order -- bitwise the same result on every
 *           rank.  It is how the multi-rank orchestration (reduce-scatter by Gaussian blocks, sharded updateTestimate,
 *           all-gather) is exercised end to end on a one-GPU machine; calls block until the exchange is complete.
 * gmmiv_comm_get_unique_id_for(transport, id): transport "rccl", "shm", or NULL = the environment variable
 * GMMIV_COMM_TRANSPORT, else "rccl"; gmmiv_comm_get_unique_id(id) = gmmiv_comm_get_unique_id_for(NULL, id).
 * A rank that waits longer than GMMIV_COMM_TIMEOUT_S (default 300) for its peers in the shm transport fails with GMMIV_ERR_HIP. */
typedef struct gmmiv_comm gmmiv_comm;
#define GMMIV_COMM_ID_BYTES 128
int gmmiv_comm_get_unique_id(void *id128);
int gmmiv_comm_get_unique_id_for(const char *transport, void *id128);
/* rank 0: creates the id and publishes it at `path` (atomically; a file already there is removed first); other ranks: wait up
 * to timeout_s for it and read it (a file last modified more than 10 minutes before the call is taken for a dead job's leftover
 * and ignored).  Rank 0 removes the file again as soon as its gmmiv_comm_create on that id has succeeded -- every rank has read
 * the id by then -- so a path can be reused by the next job. */
int gmmiv_comm_exchange_id_file(const char *path, int rank, void *id128, double timeout_s);
int gmmiv_comm_create(gmmiv_ctx *ctx, int world, int rank, const void *id128, gmmiv_comm **out);
void gmmiv_comm_destroy(gmmiv_comm *comm);
int gmmiv_comm_world(const gmmiv_comm *comm);
int gmmiv_comm_rank(const gmmiv_comm *comm);
const char *gmmiv_comm_backend(const gmmiv_comm *comm); /* "rccl: <path of the library in use>", "shm (...)", or "single rank ..." */
/* What the collective library itself says about this communicator: *rccl_version = ncclGetVersion's code (e.g. 22105), *rccl_comm_count
 * = ncclCommCount(comm), the number of ranks RCCL sees.  Both 0 for a single-rank or "shm" communicator (no RCCL behind it). */
int gmmiv_comm_info(const gmmiv_comm *comm, int *rccl_version, int *rccl_comm_count);
/* payload bytes this rank passed to collectives since the last call of this function (then reset to 0) */
double gmmiv_comm_take_bytes(gmmiv_comm *comm);
/* buf[n] <- sum over ranks (in place; host or device) */
int gmmiv_allreduce_f64(gmmiv_comm *comm, double *buf, size_t n);
/* recv[recvcount] <- block `rank` of the sum over ranks of send[world * recvcount] (device; in place when
 * recv == send + rank * recvcount) */
int gmmiv_reduce_scatter_f64(gmmiv_comm *comm, const double *send, double *recv, size_t recvcount);
/* recv[world * sendcount] <- the ranks' send[sendcount], in rank order (device; in place when send == recv + rank * sendcount) */
int gmmiv_allgather_f64(gmmiv_comm *comm, const double *send, double *recv, size_t sendcount);
/* buf[n] on every rank <- buf of rank `root` (host or device) */
int gmmiv_broadcast_f64(gmmiv_comm *comm, double *buf, size_t n, int root);
/* Overlapped forms (device buffers): *_begin orders the collective behind everything enqueued on the context's stream SO FAR and
 * runs it on the communicator's own side stream -- work enqueued on the context's stream afterwards overlaps with it;
 * gmmiv_comm_join makes the context's stream wait for every collective begun since the last join (their results may be used
 * from then on; the buffers must not be touched in between).  Same arithmetic as the plain calls: results are bitwise equal.
 * All ranks must issue begins, joins and plain collectives of one communicator in the same order.  With one rank, and on the
 * "shm" transport (whose calls block), *_begin behaves like the plain call and the join is a no-op. */
int gmmiv_allreduce_f64_begin(gmmiv_comm *comm, double *buf, size_t n);
int gmmiv_reduce_scatter_f64_begin(gmmiv_comm *comm, const double *send, double *recv, size_t recvcount);
int gmmiv_allgather_f64_begin(gmmiv_comm *comm, const double *send, double *recv, size_t sendcount);
int gmmiv_comm_join(gmmiv_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* GMMIV_H */
