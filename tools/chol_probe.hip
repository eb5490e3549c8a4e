// chol_probe.hip -- times the three one-workgroup-per-system kernels of chol_fused.hip (k_chol_left, k_trinv_left, k_uut) on nb SPD
// systems of order n as the T-matrix E-step hands them over (packed lower rows + I), checks the result against a host reference for
// one system, and -- built with -DCHOL_PROF -- prints the per-panel phase time stamps of workgroup 0 (s_memtime, waves 0 and 1).
//   build: bash tools/chol_probe.sh build      run (GPU box): tools/bin/chol_probe [n] [nb] [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../lia_ral_amd/csrc/tv_kernels.h"

#define CK(x) do { hipError_t e_ = (hipError_t)(x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)

extern long long *g_chol_prof; // chol_fused.hip, -DCHOL_PROF builds: device buffer [3 kernels][2 waves][32 panels][16 stamps]

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 400, nb = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 5;
    static gmmiv_kopts ko; // the launchers read the options bound to the thread: CHOL_WAVES / CHOL_FLOW / CHOL_LDS select the variants
    if (getenv("CHOL_WAVES")) ko.chol_waves = atoi(getenv("CHOL_WAVES"));
    if (getenv("CHOL_FLOW")) ko.chol_flow = atoi(getenv("CHOL_FLOW"));
    if (getenv("CHOL_LDS")) ko.chol_lds = atoi(getenv("CHOL_LDS"));
    gmmiv_kopts_bind(&ko);
    const long P = (long)n * (n + 1) / 2, nn = (long)n * n, nblk = (n + 31) / 32;
    std::vector<double> hp((size_t)nb * P), haux((size_t)nb * n);
    srand(7);
    // L_u - I = sum_c N_uc TETt_c: SPD, diagonally heavy like the real systems (B B^T / n + a little)
    std::vector<double> B((size_t)n * 24);
    for (int b = 0; b < nb; ++b) {
        if (b < 4) {
            for (auto &v : B) v = (double)rand() / RAND_MAX - 0.5;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s = 0;
                    for (int k = 0; k < 24; ++k) s += B[(size_t)i * 24 + k] * B[(size_t)j * 24 + k];
                    hp[(size_t)b * P + (long)i * (i + 1) / 2 + j] = s * 3.0 + (i == j ? 0.5 : 0.0);
                }
        } else memcpy(&hp[(size_t)b * P], &hp[(size_t)(b & 3) * P], P * sizeof(double));
        for (int i = 0; i < n; ++i) haux[(size_t)b * n + i] = (double)rand() / RAND_MAX - 0.5;
    }
    double *dP, *dP0, *Lf, *U, *invd, *aux, *W;
    int *status;
    CK(hipMalloc(&dP, (size_t)nb * P * 8)); CK(hipMalloc(&dP0, (size_t)nb * P * 8));
    CK(hipMalloc(&Lf, (size_t)nb * nn * 8)); CK(hipMalloc(&U, (size_t)nb * nn * 8));
    CK(hipMalloc(&invd, (size_t)nb * nblk * 1024 * 8)); CK(hipMalloc(&aux, (size_t)nb * n * 8)); CK(hipMalloc(&W, (size_t)nb * n * 8));
    CK(hipMalloc(&status, nb * sizeof(int)));
    CK(hipMemcpy(dP0, hp.data(), (size_t)nb * P * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(aux, haux.data(), (size_t)nb * n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(status, 0, nb * sizeof(int))); CK(hipMemset(Lf, 0, (size_t)nb * nn * 8)); CK(hipMemset(U, 0, (size_t)nb * nn * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e[5];
    for (auto &x : e) CK(hipEventCreate(&x));
    double ms[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemcpyAsync(dP, dP0, (size_t)nb * P * 8, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e[0], st));
        CK(tvk_chol_left_batched(st, n, nb, Lf, invd, status, dP, P, 1.0));
        CK(hipEventRecord(e[1], st));
        CK(tvk_chol_solve_batched(st, n, nb, Lf, invd, aux, W));
        CK(hipEventRecord(e[2], st));
        CK(tvk_trinv_left_batched(st, n, nb, Lf, invd, U));
        CK(hipEventRecord(e[3], st));
        CK(tvk_uut_packed_batched(st, n, nb, U, W, dP, P));
        CK(hipEventRecord(e[4], st));
        CK(hipStreamSynchronize(st));
        if (r == 0) continue; // warm-up
        for (int k = 0; k < 4; ++k) { float t; CK(hipEventElapsedTime(&t, e[k], e[k + 1])); ms[k] += t / reps; }
    }
    const double gf = (double)n * n * n / 3.0 * nb / 1e9;
    printf("n %d nb %d: k_chol_left %.3f ms (%.1f TF)  k_chol_solve %.3f  k_trinv_left %.3f ms (%.1f TF)  k_uut %.3f ms (%.1f TF)   sum of three %.3f ms\n", n, nb,
           ms[0], gf / ms[0], ms[1], ms[2], gf / ms[2], ms[3], gf / ms[3], ms[0] + ms[2] + ms[3]);
    // check system 1: E = (P + I)^-1 + w w^T against a host Cholesky
    {
        const int b = 1 % nb;
        std::vector<double> A((size_t)nn), E((size_t)P), w(n), hW(n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = hp[(size_t)b * P + (long)i * (i + 1) / 2 + j] + (i == j ? 1.0 : 0.0);
        std::vector<double> L(A);
        for (int j = 0; j < n; ++j) {
            for (int k = 0; k < j; ++k)
                for (int i = j; i < n; ++i) L[(size_t)i * n + j] -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            const double d = sqrt(L[(size_t)j * n + j]);
            for (int i = j; i < n; ++i) L[(size_t)i * n + j] /= d;
        }
        std::vector<double> inv((size_t)nn, 0.0), col(n);
        for (int c = 0; c < n; ++c) { // solve A x = e_c
            for (int i = 0; i < n; ++i) { double s = (i == c); for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * col[k]; col[i] = s / L[(size_t)i * n + i]; }
            for (int i = n - 1; i >= 0; --i) { double s = col[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * col[k]; col[i] = s / L[(size_t)i * n + i]; }
            for (int i = 0; i < n; ++i) inv[(size_t)i * n + c] = col[i];
        }
        for (int i = 0; i < n; ++i) { double s = 0; for (int k = 0; k < n; ++k) s += inv[(size_t)i * n + k] * haux[(size_t)b * n + k]; w[i] = s; }
        CK(hipMemcpy(E.data(), dP + (size_t)b * P, P * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hW.data(), W + (size_t)b * n, n * 8, hipMemcpyDeviceToHost));
        double ee = 0, em = 0, we = 0, wm = 0;
        for (int i = 0; i < n; ++i) {
            we = fmax(we, fabs(hW[i] - w[i])); wm = fmax(wm, fabs(w[i]));
            for (int j = 0; j <= i; ++j) { const double ref = inv[(size_t)i * n + j] + w[i] * w[j]; ee = fmax(ee, fabs(E[(long)i * (i + 1) / 2 + j] - ref)); em = fmax(em, fabs(ref)); }
        }
        int hs = 0;
        CK(hipMemcpy(&hs, status + b, sizeof(int), hipMemcpyDeviceToHost));
        printf("check (system %d): status %d, max rel err E %.2e, w %.2e\n", b, hs, ee / em, we / wm);
    }
    if (g_chol_prof) {
        std::vector<long long> h(3 * 2 * 32 * 16);
        CK(hipMemcpy(h.data(), g_chol_prof, h.size() * 8, hipMemcpyDeviceToHost));
        const char *names[3] = {"k_chol_left", "k_trinv_left", "k_uut"};
        for (int k = 0; k < 3; ++k)
            for (int wv = 0; wv < 2; ++wv) {
                printf("%s wave %d: per panel, cycles between stamps (s_memtime)\n", names[k], wv);
                long long tot[16] = {0};
                for (int p = 0; p < nblk && p < 32; ++p) {
                    const long long *s = &h[((k * 2 + wv) * 32 + p) * 16];
                    printf("  panel %2d:", p);
                    for (int i = 1; i < 16 && s[i]; ++i) { printf(" %7lld", s[i] - s[i - 1]); tot[i] += s[i] - s[i - 1]; }
                    printf("\n");
                }
                printf("  total   :");
                for (int i = 1; i < 16 && tot[i]; ++i) printf(" %7lld", tot[i]);
                printf("\n");
            }
    }
    return 0;
}
