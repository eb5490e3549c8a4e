// mfma_power_probe.hip -- does the sustained rate of v_mfma_f64_16x16x4_f64 on an MI355X depend on the DATA?  (round 6: k_dgemm runs the
// same 4096^3 product at 69.5 TF on zero-filled operands and at 61.5 TF on pseudo-random ones -- tools/gemm_probe with GEMM_FILL=-1.)
// Pure register kernel, no memory traffic in the loop: 8 independent accumulators per wave, 8 waves per CU x 256 CUs x 2 workgroups,
// operands either one constant per lane (MODE 0) or a rotating set of 16 pseudo-random values per lane (MODE 1: mantissa bits toggle on
// every issue), accumulators kept bounded.  Prints TFLOP/s over a run long enough for the power management to settle (~0.3 s per mode).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_power_probe tools/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_mfma(double *out, int iters, unsigned seed)
{
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a[16], b[16];
    unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + seed;
    for (int i = 0; i < 16; ++i) {
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = MODE ? ((double)h / 4294967296.0 - 0.5) * 1e-3 : 1e-3;
        h = h * 1664525u + 1013904223u;
        b[i] = MODE ? ((double)h / 4294967296.0 - 0.5) * 1e-3 : 0.5e-3;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = MFMA_F64(a[(k + i) & 15], b[(k + 3 * i) & 15], acc[i]);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int blocks = 512, iters = argc > 1 ? atoi(argv[1]) : 40000;
    double *out;
    CK(hipMalloc(&out, (size_t)blocks * 512 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipEventRecord(e0, 0));
            if (mode) k_mfma<1><<<blocks, 512>>>(out, iters, 7u); else k_mfma<0><<<blocks, 512>>>(out, iters, 7u);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)blocks * 8 /* waves */ * iters * 16.0 * 8.0 * 2048.0;
            printf("%-28s %8.1f ms  %6.1f TFLOP/s\n", mode ? "pseudo-random operands" : "one constant per lane", ms, flop / ms / 1e9);
        }
    return 0;
}
