// tv_kernels.h -- launchers of tv_kernels.hip (internal). Device pointers only; return hipError_t values.
#pragma once
#include <hip/hip_runtime.h>

#include "kopts.h"

int tvk_dgemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda, long sA,
              const double *B, long ldb, long sB, double beta, double *C, long ldc, long sC, int batch);
int tvk_chol_batched(hipStream_t st, int n, int nb, double *Afull, double *invd, double *panel, int *status);
int tvk_chol_left_batched(hipStream_t st, int n, int nb, double *Afull, double *invd, int *status, const double *Apacked = nullptr,
                          long spk = 0, double diag_add = 0.0);
int tvk_chol_solve_multi_batched(hipStream_t st, int n, int nb, int nrhs, const double *Lf, const double *invd, const double *B, long ldb,
                                 long sB, double *X, long ldx, long sX); // -1: nrhs > 64 or odd n (use the explicit inverse)
int tvk_trinv_left_batched(hipStream_t st, int n, int nb, const double *Lf, const double *invd, double *U);      // U = L^-T (chol_fused.hip)
int tvk_uut_packed_batched(hipStream_t st, int n, int nb, const double *U, const double *w, double *P, long sp); // P = packed(U U^T + w w^T)
int tvk_subtract_m_to(hipStream_t st, long U, int C, int D, const double *N, const double *Fs, double *Fd, const double *means); // -1: odd D / unaligned
int tvk_chol_accepts_packed(int n); // 1 when the batched factorisation can read packed lower rows directly
// (the A/B switches "chol_gemm", "chol_lds", "gemm_clamp", "gemm_narrow", "gemm_remap" are options of the calling context:
//  ctx.h, gmmiv_kopts -- the launchers read gmmiv_kopts_cur())
int tvk_spd_inverse_left_batched(hipStream_t st, int n, int nb, double *Afull, double *inv, double *U, double *invd, int *status,
                                 const double *Apacked = nullptr, long spk = 0, double diag_add = 0.0);
int tvk_spd_inverse_batched(hipStream_t st, int n, int nb, double *Afull, double *inv, double *X, double *invd,
                            double *panel, int *status);
int tvk_subtract_m(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means);
int tvk_scale_cols(hipStream_t st, long rows, long cols, const double *in, const double *scale, double *out);
int tvk_unpack_sym(hipStream_t st, int n, int nb, const double *packed, long sp, double *full, double diag_add);
int tvk_tett_packed(hipStream_t st, int C, int D, int R, const double *T, const double *iv, double *out); // -1: shape not served (D > 64)
int tvk_pack_sym(hipStream_t st, int n, int nb, const double *full, long sf, const double *w, double *packed, long sp);
int tvk_merge_rows(hipStream_t st, long ndst, long width, const long *off, const long *rows, const double *src, double *dst);
#define TVK_BATCH_SUM_SLABS 16
int tvk_batch_sum(hipStream_t st, long n, int nb, const double *src, long stride, double *dst, double *tmp = nullptr); // tmp: SLABS * n doubles
#define TVK_NARROW_SLABS_DOUBLES(n) ((size_t)32 * (size_t)(n))
int tvk_colsum_narrow(hipStream_t st, int n, int nb, const double *src, long stride, double *dst, double *dst2, double *tmp); // tmp: 32 * n doubles
int tvk_add_unpacked(hipStream_t st, int n, const double *packed, double *full);
int tvk_md_normalize(hipStream_t st, int R, double n_sessions, double *Rm, double *r, double *work); // Rm, work <- Rm / n - (r / n)(r / n)^T ; r /= n
int tvk_lower_to_upper(hipStream_t st, int n, const double *L, double *U);                       // U = L^T, zero below the diagonal
int tvk_batched_matvec(hipStream_t st, int n, int nb, const double *Mx, const double *x, double *y);
int tvk_vecmat_add(hipStream_t st, int rows, long cols, const double *x, const double *Mx, double *y);
int tvk_coldot(hipStream_t st, int dim, long n, const double *X, const double *Y, double *qv, long ld = 0); // ld: row stride of X and Y (0: n)
int tvk_add_transpose(hipStream_t st, int n, const double *a, const double *b, double *out);
int tvk_axpby(hipStream_t st, long n, double a, const double *x, double b, const double *y, double *out);
int tvk_splitk_count(int M, int N, int K, int n_cu);
int tvk_dgemm_splitk(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda,
                     const double *B, long ldb, double beta, double *C, long ldc, int nz, double *slabs);
int tvk_sub_colvec(hipStream_t st, int dim, long n, const double *X, const double *v, double *out);
int tvk_scale_cols_rsqrt(hipStream_t st, int dim, long n, double *X, const double *qv);
int tvk_orthonormalize(hipStream_t st, int R, long SV, const double *Tm, double *Q, double *rv, double *v, double *partial);
int tvk_chol_solve_batched(hipStream_t st, int n, int nb, const double *Lf, const double *invd, const double *b, double *w);
int tvk_norm_stats(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means, const double *invvar);
int tvk_sub_mtw(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means, const double *TW);
int tvk_scale_cols_fn(hipStream_t st, long rows, long cols, int D, int mode, const double *in, const double *v, double *out);
int tvk_block_colnorm(hipStream_t st, int C, int D, int R, const double *A, double *Dm);
int tvk_build_l_ubm(hipStream_t st, int R, int C, int nb, const double *N, const double *Wm, double *full);
int tvk_mul_recip1p(hipStream_t st, long n, double *B, const double *X);
int tvk_add_identity(hipStream_t st, int n, double *A);
int tvk_dev_means(hipStream_t st, int dim, long n, const double *X, long nspk, const long *off, double *ssum, double *mean, double *smean);
int tvk_dev_center(hipStream_t st, int dim, long n, int mode, const double *X, const double *mean, const double *smean, long nspk,
                   const long *off, const int *cls, double *out);
int tvk_dev_between(hipStream_t st, int dim, long nspk, int weighted, const double *mean, const double *smean, const long *off, double *out);
int tvk_dev_expand(hipStream_t st, int rows, long n, long nspk, const double *H, const int *cls, double *out);
int tvk_gather_rows(hipStream_t st, int nb, int R, long r0, const long *owner, const double *W, double *Wg);
int tvk_jfa_sub(hipStream_t st, long nb, int C, int D, long r0, const long *owner, const double *N, double *F, const double *means,
                const double *TW, const double *Dm, const double *Z);
int tvk_jfa_sub_sessions(hipStream_t st, long s0, long ns, long h0, long h1, int C, int D, const long *sess_begin, const double *Nh,
                         const double *G, double *FX);
int tvk_jfa_z(hipStream_t st, long nspk, int C, int D, const double *N, const double *F, const double *iv, const double *Dm, double tau, double *Z);
int tvk_jfa_z_and_d(hipStream_t st, long nspk, int C, int D, const double *N, const double *F, const double *iv, double *Dm, double *Z);
int tvk_dgemm_epi(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda, const double *B,
                  long ldb, double *C, long ldc, int mode, const double *rv, const double *cv, double br, double bc, double cst, double beta = 0.0);
int tvk_mask_trials(hipStream_t st, long n, const unsigned char *trials, double fill, double *scores);
int tvk_rsqrt_vec(hipStream_t st, long n, double *v);
int tvk_inverse_e_packed_batched(hipStream_t st, int n, int nb, double *Lf, double *U, double *invd, int *status, double *P, long sp,
                                 double diag_add, const double *aux, double *W);
