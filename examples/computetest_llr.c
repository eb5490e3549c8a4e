/* computetest_llr.c -- ComputeTest's score for one segment through the C ABI alone (C99, no HIP / C++ / Python on the caller's side):
 * world model + client model + float32 frames in, LLR = mean llk_client - mean llk_world out, with the world's top-C selection
 * re-used for the client (LIA_SpkDet/ComputeTest/src/ComputeTest.cpp:154-207).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/computetest_llr.c -Llia_ral_amd/csrc -lgmmiv -Wl,-rpath,$PWD/lia_ral_amd/csrc -lm -o computetest_llr
 *   ./computetest_llr model.bin          (model.bin: int32 C, D, T, topC; double w[C], mean_w[C*D], covinv[C*D], mean_c[C*D]; float x[T*D])
 * tests/test_gpu_kat5.py::test_c99_example_reproduces_the_reference_llr writes the file from the ComputeTest golden (KAT-1) and checks
 * the printed LLR against test1.validate.res (5.06601). */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "gmmiv.h"

#define CHECK(call) do { if ((call) != GMMIV_OK) { fprintf(stderr, "%s -> %s\n", #call, gmmiv_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s model.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hdr[4];
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
    const int C = hdr[0], D = hdr[1], T = hdr[2], topC = hdr[3];
    const size_t CD = (size_t)C * D;
    double *w = malloc(C * sizeof *w), *mw = malloc(CD * sizeof *mw), *iv = malloc(CD * sizeof *iv), *mc = malloc(CD * sizeof *mc);
    float *x = malloc((size_t)T * D * sizeof *x);
    if (fread(w, sizeof *w, C, f) != (size_t)C || fread(mw, sizeof *mw, CD, f) != CD || fread(iv, sizeof *iv, CD, f) != CD ||
        fread(mc, sizeof *mc, CD, f) != CD || fread(x, sizeof *x, (size_t)T * D, f) != (size_t)T * D) { fprintf(stderr, "short file\n"); return 2; }
    fclose(f);

    gmmiv_ctx *ctx;
    gmmiv_gmm *world, *client;
    CHECK(gmmiv_ctx_create(0, NULL, &ctx));                       /* device 0, a stream of the context's own */
    CHECK(gmmiv_gmm_create(ctx, C, D, w, mw, iv, &world));
    CHECK(gmmiv_gmm_create(ctx, C, D, w, mc, iv, &client));
    int32_t *idx = malloc((size_t)T * topC * sizeof *idx);
    double *rest = malloc(T * sizeof *rest), *llkw = malloc(T * sizeof *llkw), *llkc = malloc(T * sizeof *llkc);
    /* DETERMINE_TOP_DISTRIBS on the world model (host arrays in and out: the library stages them) ... */
    CHECK(gmmiv_llk_determine_top(ctx, world, x, GMMIV_F32, T, D, topC, GMMIV_TOP_COMPLETE, -200.0, 200.0, idx, NULL, NULL, rest, NULL, llkw));
    /* ... USE_TOP_DISTRIBS on the client */
    CHECK(gmmiv_llk_use_top(ctx, client, x, GMMIV_F32, T, D, topC, idx, rest, GMMIV_TOP_COMPLETE, -200.0, 200.0, llkc));
    double sw = 0.0, sc = 0.0;
    for (int t = 0; t < T; ++t) { sw += llkw[t]; sc += llkc[t]; }
    printf("frames %d  mean llk world %.6f  client %.6f  LLR %.6f  (%s)\n", T, sw / T, sc / T, sc / T - sw / T, gmmiv_version());
    gmmiv_gmm_destroy(client); gmmiv_gmm_destroy(world); gmmiv_ctx_destroy(ctx);
    free(w); free(mw); free(iv); free(mc); free(x); free(idx); free(rest); free(llkw); free(llkc);
    return 0;
}
