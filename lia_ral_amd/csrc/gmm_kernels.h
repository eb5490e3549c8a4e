// gmm_kernels.h -- host-callable launchers of gmm_kernels.hip (internal; the public surface is
// include/gmmiv.h).  All pointers are device pointers; every function returns a hipError_t value.
#pragma once
#include <hip/hip_runtime.h>

#include "kopts.h"

#define GMMK_KS_GENERIC 99 // "no MFMA instantiation for this vectSize": every KS > 15 / KS <= 15 test of the callers routes it to the fallback side
#define GMMK_MAX_DIM 4096  // the generic kernels keep 4 frames x D doubles in LDS
int gmmk_ks_for_dim(int D);   // k-steps (of 4 dims) of the compiled instantiation serving D, 0 = unsupported
int gmmk_rl_for_ks(int KS);   // row length (doubles) of the LDS frame tile / half-width of an EM partial row
int gmmk_pack_model(hipStream_t st, int C, int D, int KS, int nct, int Cp64, const double *w, const double *mean,
                    const double *iv, double *a, double *lwc, double *Pt, double *meanT, double *ivT);
int gmmk_llk(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct,
             double *lse, int use_glds, int wg_waves);
int gmmk_llk_finalize(hipStream_t st, const double *lse, long T, double lo, double hi, double *llk_out,
                      double *partial /* >= 3*256 doubles */, double scale_c, double scale_r, double *dst_clamped,
                      double *dst_raw, double scale_n = 0.0, double *dst_count = nullptr);
int gmmk_add_scalar(hipStream_t st, double *dst, double v);
int gmmk_count_dead(hipStream_t st, const double *lse, long T, unsigned long long *cnt); // *cnt += frames whose lse is not finite (zero-likelihood frames)
int gmmk_rows_sum_groups(hipStream_t st, long n, int ngroups, const int *rb, const double *src, double *dst); // dst[g] = sum of src rows [rb[g], rb[g+1])
int gmmk_stats(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, const double *Pt,
               int nct, const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0,
               double *out1, int mode, int wg_waves, double prune_arg);
int gmmk_em_reduce(hipStream_t st, const double *part, int nseg, int C, int Cp, int D, int KS, double *acc);
int gmmk_em_get(hipStream_t st, int C, int D, const double *acc, const double *prev_mean, const double *prev_cov,
                double *w, double *mean, double *cov);
int gmmk_topc_frames_per_block(int Cp64, int D);
int gmmk_topc_determine(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp,
                        const double *meanT, const double *ivT, const double *lwc, const double *w, int ctop,
                        int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                        double *nw, double *llk);
// generic statistics (capi_gmm.hip, models without an MFMA instantiation): Xa rows for the GEMM, scatter of S = gamma^T Xa
int gmmk_build_xa(hipStream_t st, int x_f64, const void *x, long ldx, int D, long n, int sq, int NC, double *Xa);
int gmmk_scatter_em(hipStream_t st, int C, int D, int NC, const double *S, double scale, double *acc);
int gmmk_scatter_nf(hipStream_t st, int C, int D, int NC, const double *S, double *Nrow, double *Frow);
size_t gmmk_topc_big_scratch_doubles(long T, int Cp);
int gmmk_topc_determine_big(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp,
                            const double *meanT, const double *ivT, const double *lwc, const double *w, int ctop,
                            int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                            double *nw, double *llk, double *zs); // any C / D / ctop: logit rows in global scratch (gmmk_topc_big_scratch_doubles)
int gmmk_topc_use_big(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean,
                      const double *iv, const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete,
                      double lo, double hi, double *llk); // ctop > 64
int gmmk_topc_use(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean,
                  const double *iv, const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete,
                  double lo, double hi, double *llk);
int gmmk_frame_moments(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, double *partial,
                       int max_blocks, double *acc);
int gmmk_variance_control(hipStream_t st, int C, int D, double *cov, double flooring, double ceiling,
                          const double *cov_signal, unsigned long long *counts);
int gmmk_reciprocal(hipStream_t st, long n, const double *in, double *out);
int gmmk_gather_frames(hipStream_t st, int x_f64, const void *x, long ldx, int D, const long *idx, long n, void *out);
int gmmk_segment_means(hipStream_t st, const double *v, long ld, const long *item, long nitem, double *part, const long *pair_off,
                       const long *pair_len, long npair, double *out);
int gmmk_topgauss_select(hipStream_t st, long T, int cap, double mass, int fixed_count, const double *w, int *idx, const double *lk,
                         const double *llk, int *count, double *snsw, double *snsl, unsigned long long *capped);
int gmmk_flag_frames(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, unsigned char *flag, int *any);
int gmmk_fill_chunks(hipStream_t st, long *dst, int nseg, long per, long n);
int gmmk_count_flags(hipStream_t st, const unsigned char *flag, long T, unsigned long long *cnt);
int gmmk_gather_runs(hipStream_t st, int x_f64, const void *x, long ldx, int D, const long *runs, long nrun, void *out);

// stats_z.hip / k_llk_mfma<WZ>: scaled likelihoods written once by the log-likelihood kernel, statistics from them
int gmmk_llk_z(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct,
               double *lse, int use_glds, double *zbuf, long nfb, int *eit, double *inv, int *efin);
// k_llk_mfma<TC>: candidates of the top-C' selection collected in the log-likelihood kernel (see gmm_kernels.hip), ranked by
// gmmk_topc_rank (topc_z.hip)
int gmmk_topc_cap(void);
int gmmk_llk_topc(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct, int use_glds,
                  int ctop, double *cand, int *cnt, double *theta, double *slow, int *efin);
int gmmk_topc_rank(hipStream_t st, int x_f64, const void *x, long n, long ldx, int D, int C, const double *cand, const int *cnt,
                   const double *theta, const double *slow, const int *efin, const double *mean, const double *iv, const double *lwc,
                   const double *w, int ctop, int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                   double *nw, double *llk, int *flag, long *redo, int stats, long *wide);
int gmmk_topc_scatter(hipStream_t st, long n, int ctop, const long *redo, const int *sidx, const double *slk, const double *snlk,
                      const double *snllk, const double *snw, const double *sllk, int *idx, double *lk, double *nlk, double *nllk, double *nw,
                      double *llk);
int gmmk_posteriors(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp, const double *meanT,
                    const double *ivT, const double *lwc, const double *lse, double *gamma);
int gmmk_stats_z_groups(int nct);
int gmmk_stats_z_wg_per_cu(void);
int gmmk_stats_z(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, int nct, const double *zbuf,
                 long nfb, const int *eit, const double *inv, const int *efin, double scale, const long *seg_begin, int nseg,
                 double *out0, double *out1, int mode, int accum, double prune_thr);
size_t gmmk_topc_z_lds(int nct, int D);
int gmmk_topc_from_z(hipStream_t st, int x_f64, const void *x, long n, long ldx, int D, int C, int nct, const double *zbuf, long nfb,
                     const int *eit, const int *efin, const double *mean, const double *iv, const double *lwc,
                     const double *w, int ctop, int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                     double *nw, double *llk, int *flag);
int gmmk_topc_use16(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean, const double *iv,
                    const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete, double lo, double hi, double *llk, int four);
int gmmk_topc_use4_multi(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const void *clients, int n_clients, int ctop,
                         const int *idx, const double *nllk, int complete, double lo, double hi, double *llk); // clients: device array of {mean, iv, lwc, (long) C}
int gmmk_post_from_z(hipStream_t st, long n, int C, int nct, const double *zbuf, long nfb, const int *eit, const double *inv,
                     const int *efin, double *gamma);
