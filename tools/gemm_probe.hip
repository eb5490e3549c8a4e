// gemm_probe.hip -- times tvk_dgemm (lia_ral_amd/csrc/tv_kernels.hip) on the shapes the i-vector path uses.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip lia_ral_amd/csrc/chol_fused.hip -o tools/bin/gemm_probe
#include "../lia_ral_amd/csrc/tv_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k_fill_hash(double *p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((double)h / 4294967296.0 - 0.5) * 2.0;      // uniform in (-1, 1): every mantissa bit toggles
    }
}

struct Shape { const char *name; bool ta, tb; int M, N, K; double beta = 0.0; };

int main()
{
    const Shape shapes[] = {
        {"square 4096^3 NN", false, false, 4096, 4096, 4096},
        {"square 4096^3 NT", false, true, 4096, 4096, 4096},
        {"L = N TETt: 256 x 80200 x 2048 NN", false, false, 256, 80200, 2048},
        {"A += N^T E: 2048 x 80200 x 256 TN", true, false, 2048, 80200, 256},
        {"Cmx += W^T F: 400 x 122880 x 256 TN", true, false, 400, 122880, 256},
        {"scores: 20000 x 20000 x 400 TN", true, false, 20000, 20000, 400},
        {"scores: 20000 x 20000 x 200 TN", true, false, 20000, 20000, 200},
        {"scores: 20000 x 20000 x 16 TN (store-bound probe)", true, false, 20000, 20000, 16},
        {"aux = F Tiv^T: 1024 x 400 x 122880 NT (one K layer; the product splits K)", false, true, 1024, 400, 122880},
        {"L = N TETt: 1024 x 80200 x 2048 NN", false, false, 1024, 80200, 2048},
        {"L, N a multiple of 128: 1024 x 80128 x 2048 NN", false, false, 1024, 80128, 2048},
        {"A += N^T E: 2048 x 80200 x 1024 TN beta 1", true, false, 2048, 80200, 1024, 1.0},
        {"A  = N^T E: 2048 x 80200 x 1024 TN beta 0", true, false, 2048, 80200, 1024},
        {"A  = N E (NN): 2048 x 80200 x 1024 NN beta 0", false, false, 2048, 80200, 1024},
        {"Cmx += W^T F: 400 x 122880 x 1024 TN beta 1", true, false, 400, 122880, 1024, 1.0},
        {"Cmx, 384 rows: 384 x 122880 x 1024 TN beta 1", true, false, 384, 122880, 1024, 1.0},
        {"Cmx += W^T F: 400 x 122880 x 6250 TN beta 1", true, false, 400, 122880, 6250, 1.0},
        {"Cmx, 384 rows: 384 x 122880 x 6250 TN beta 1", true, false, 384, 122880, 6250, 1.0},
        {"A += N^T E: 2048 x 80200 x 6250 TN beta 1", true, false, 2048, 80200, 6250, 1.0},
        {"4096 x 4096 x 1024 NN", false, false, 4096, 4096, 1024},
        {"4096 x 4096 x 2048 NN", false, false, 4096, 4096, 2048},
        {"8192 x 8192 x 2048 NN", false, false, 8192, 8192, 2048},
        {"T_c = A_c^-1 C_c: 400 x 60 x 400 NN", false, false, 400, 60, 400},
    };
    hipStream_t st;
    hipStreamCreate(&st);
    const char *only = getenv("GEMM_ONLY");   // substring of the shape names to run (PMC passes want one kernel shape per run)
    for (const Shape &s : shapes) {
        if (only && !strstr(s.name, only)) continue;
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.K * s.N, nc = (size_t)s.M * s.N;
        double *A, *B, *C;
        hipMalloc(&A, na * 8); hipMalloc(&B, nb * 8); hipMalloc(&C, nc * 8);
        const int fill = getenv("GEMM_FILL") ? atoi(getenv("GEMM_FILL")) : 0;   // byte pattern of the operands: 0 = zeros; 63 (0x3f) = 4.8e-4 in every element
        hipMemset(A, fill, na * 8); hipMemset(B, fill, nb * 8); hipMemset(C, 0, nc * 8);
        if (fill < 0) { k_fill_hash<<<4096, 256, 0, st>>>(A, na, 1u); k_fill_hash<<<4096, 256, 0, st>>>(B, nb, 2u); }   // GEMM_FILL=-1: pseudo-random operands
        const long lda = s.ta ? s.M : s.K, ldb = s.tb ? s.K : s.N;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        tvk_dgemm(st, s.ta, s.tb, s.M, s.N, s.K, 1.0, A, lda, 0, B, ldb, 0, s.beta, C, s.N, 0, 1);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        const int reps = 3;
        for (int r = 0; r < reps; ++r) tvk_dgemm(st, s.ta, s.tb, s.M, s.N, s.K, 1.0, A, lda, 0, B, ldb, 0, s.beta, C, s.N, 0, 1);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("%-75s %8.3f ms  %6.1f TFLOP/s\n", s.name, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        hipFree(A); hipFree(B); hipFree(C);
    }
    if (!only) { // split-K of aux = F Tiv^T (1024 x 400 x 122880 NT): which layer count fills 2 x 256 workgroup slots best?
        const int M = 1024, N = 400, K = 122880;
        double *A, *B, *C, *slabs;
        hipMalloc(&A, (size_t)M * K * 8); hipMalloc(&B, (size_t)N * K * 8); hipMalloc(&C, (size_t)M * N * 8); hipMalloc(&slabs, (size_t)128 * M * N * 8);
        hipMemset(A, 0, (size_t)M * K * 8); hipMemset(B, 0, (size_t)N * K * 8);
        for (int nz : {8, 12, 16, 21, 24, 32, 42, 48, 64, 96}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            tvk_dgemm_splitk(st, false, true, M, N, K, 1.0, A, K, B, K, 0.0, C, N, nz, slabs);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int r = 0; r < 3; ++r) tvk_dgemm_splitk(st, false, true, M, N, K, 1.0, A, K, B, K, 0.0, C, N, nz, slabs);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            printf("aux split-K, %3d layers (default %d)   %8.3f ms  %6.1f TFLOP/s\n", nz, tvk_splitk_count(M, N, K, 256), ms, 2.0 * M * N * K / ms / 1e9);
        }
    }
    return 0;
}
