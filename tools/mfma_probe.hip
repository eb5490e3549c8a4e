// mfma_probe.hip -- gfx950 micro-benchmarks that the kernel design rests on (run on the GPU box):
//   1. lane layout of v_mfma_f64_16x16x4_f64 (A, B, C/D maps assumed by gmm_kernels.hip)
//   2. issue rate of the f64 MFMA, of v_fma_f64, and of the software exp, alone and mixed
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../lia_ral_amd/csrc/devutil.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

__global__ void k_layout(const double *A /*16x4*/, const double *B /*4x16*/, double *Dm /*16x16*/)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = MFMA_F64(a, b, acc);
    for (int r = 0; r < 4; ++r) Dm[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int MODE>  // 0: mfma only, 1: fma only, 2: exp only, 3: mfma + exp interleaved in one wave
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, double seed)
{
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = -1.0 - 0.01 * i - 1e-6 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = MFMA_F64(a, b, acc[i]);
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], 0.999999, 1e-9);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gexp(v[i]) - 1.5;
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// half the waves of a block run MFMA, the other half exp: do the pipes overlap?
__global__ __launch_bounds__(512) void k_mixed(double *out, int iters, double seed)
{
    const int wave = threadIdx.x >> 6;
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = -1.0 - 0.01 * i - 1e-6 * threadIdx.x;
    if (wave & 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gexp(v[i]) - 1.5;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_n(double *out, int iters, double seed)
{
    d4 acc[NACC];
    double a[NACC], b[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (d4){seed, seed, seed, seed}; a[i] = seed + i + threadIdx.x * 1e-9; b[i] = seed * 0.5 - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = MFMA_F64(a[i], b[i], acc[i]);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same sweep with the MFMA pinned to VGPR accumulators by inline asm (hipcc otherwise shuttles the
// accumulators through AGPRs inside the loop of the plain-builtin probe above)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_asm(double *out, int iters, double seed)
{
    d4 acc[NACC];
    double a[NACC], b[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (d4){seed, seed, seed, seed}; a[i] = seed + i + threadIdx.x * 1e-9; b[i] = seed * 0.5 - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave: each MFMA is followed by NV independent v_fma_f64 -- how many hide under a 64-cycle MFMA?
template <int NV>
__global__ __launch_bounds__(256) void k_mfma_valu(double *out, int iters, double seed)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0 + 0.01 * i + 1e-6 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(b), "v"(a));
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same with f32 FMAs / integer adds: do 32-bit VALU ops hide under the fp64 MFMA?
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_mfma_valu32(double *out, int iters, double seed)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    float v[16]; int u[16];
    for (int i = 0; i < 16; ++i) { v[i] = 1.0f + 0.01f * i; u[i] = i + threadIdx.x; }
    const float fb = 0.999f, fa = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(fb), "v"(fa));
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(it));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 16; ++i) s += v[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> static float time_ms(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, CUs %d, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    // ---- layout
    std::vector<double> hA(64), hB(64), hD(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 0.1 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 2 + 0.01 * j * j - 0.3 * k;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += hA[i * 4 + k] * hB[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, hA.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), 64 * 8, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(hD.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int e = 0; e < 256; ++e) err = fmax(err, fabs(hD[e] - ref[e]));
    printf("LAYOUT f64 16x16x4: max |D - ref| = %.3e  -> %s\n", err, err < 1e-9 ? "OK" : "MISMATCH");
    // ---- rates
    const int blocks = p.multiProcessorCount * 2, iters = 20000;
    double *out;
    CK(hipMalloc(&out, (size_t)blocks * 512 * 8));
    const double waves = (double)blocks * 4;
    float ms = time_ms([&] { k_rate<0><<<blocks, 256>>>(out, iters, 1e-3); });
    double nm = waves * iters * 8.0;
    printf("MFMA f64 only   : %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz, 2 waves/SIMD)\n", ms, nm * 2048 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 8.0 * 2));
    ms = time_ms([&] { k_rate<1><<<blocks, 256>>>(out, iters, 1e-3); });
    printf("v_fma_f64 only  : %.3f ms  %.1f TFLOP/s\n", ms, waves * iters * 32.0 * 64 * 2 / ms / 1e9);
    ms = time_ms([&] { k_rate<2><<<blocks, 256>>>(out, iters, 1e-3); });
    printf("gexp only       : %.3f ms  %.2f Gexp/s  (%.1f cycles/wave-exp @2.4GHz per SIMD)\n", ms, waves * iters * 8.0 * 64 / ms / 1e6,
           ms * 1e-3 * 2.4e9 / (iters * 8.0 * 2));
    ms = time_ms([&] { k_rate<3><<<blocks, 256>>>(out, iters, 1e-3); });
    printf("MFMA+gexp 1 wave: %.3f ms  (same counts as the two lines above, interleaved in each wave)\n", ms);
    float ms2 = time_ms([&] { k_mixed<<<p.multiProcessorCount, 512>>>(out, iters, 1e-3); });
    printf("MFMA || gexp    : %.3f ms  (4 MFMA waves + 4 exp waves per CU, %d iters each)\n", ms2, iters);
    float ms3 = time_ms([&] { k_rate<0><<<p.multiProcessorCount, 256>>>(out, iters, 1e-3); });
    float ms4 = time_ms([&] { k_rate<2><<<p.multiProcessorCount, 256>>>(out, iters, 1e-3); });
    printf("  reference: 4 MFMA waves/CU alone %.3f ms, 4 exp waves/CU alone %.3f ms\n", ms3, ms4);
    // ---- MFMA f64 issue-rate sweep: waves per SIMD x independent accumulators
    {
        const int it2 = 40000;
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int nblk = p.multiProcessorCount * wps;   // 256-thread blocks: 1 wave per SIMD each
            float t1 = time_ms([&] { k_mfma_n<1><<<nblk, 256>>>(out, it2, 1e-3); });
            float t2 = time_ms([&] { k_mfma_n<2><<<nblk, 256>>>(out, it2, 1e-3); });
            float t4 = time_ms([&] { k_mfma_n<4><<<nblk, 256>>>(out, it2, 1e-3); });
            float t8 = time_ms([&] { k_mfma_n<8><<<nblk, 256>>>(out, it2, 1e-3); });
            const double fl = (double)nblk * 4 * it2 * 2048.0;
            printf("MFMA f64 sweep, %d wave(s)/SIMD: 1 acc %.1f TF | 2 acc %.1f TF | 4 acc %.1f TF | 8 acc %.1f TF\n", wps,
                   fl * 1 / t1 / 1e9, fl * 2 / t2 / 1e9, fl * 4 / t4 / 1e9, fl * 8 / t8 / 1e9);
        }
    }
    {
        const int it2 = 20000, nblk = p.multiProcessorCount * 2;
        float t0 = time_ms([&] { k_mfma_valu<0><<<nblk, 256>>>(out, it2, 1e-3); });
        float t4 = time_ms([&] { k_mfma_valu<4><<<nblk, 256>>>(out, it2, 1e-3); });
        float t8 = time_ms([&] { k_mfma_valu<8><<<nblk, 256>>>(out, it2, 1e-3); });
        float t16 = time_ms([&] { k_mfma_valu<16><<<nblk, 256>>>(out, it2, 1e-3); });
        printf("MFMA + N x v_fma_f64 per MFMA in one wave (2 waves/SIMD): N=0 %.3f ms | N=4 %.3f | N=8 %.3f | N=16 %.3f\n", t0, t4, t8, t16);
        float f8 = time_ms([&] { k_mfma_valu32<8, 0><<<nblk, 256>>>(out, it2, 1e-3); });
        float f16 = time_ms([&] { k_mfma_valu32<16, 0><<<nblk, 256>>>(out, it2, 1e-3); });
        float i8 = time_ms([&] { k_mfma_valu32<8, 1><<<nblk, 256>>>(out, it2, 1e-3); });
        float i16 = time_ms([&] { k_mfma_valu32<16, 1><<<nblk, 256>>>(out, it2, 1e-3); });
        printf("MFMA + N x v_fma_f32: N=8 %.3f ms | N=16 %.3f ;  MFMA + N x v_add_u32: N=8 %.3f ms | N=16 %.3f\n", f8, f16, i8, i16);
    }
    {
        const int it2 = 40000;
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int nblk = p.multiProcessorCount * wps;
            float t1 = time_ms([&] { k_mfma_asm<1><<<nblk, 256>>>(out, it2, 1e-3); });
            float t2 = time_ms([&] { k_mfma_asm<2><<<nblk, 256>>>(out, it2, 1e-3); });
            float t4 = time_ms([&] { k_mfma_asm<4><<<nblk, 256>>>(out, it2, 1e-3); });
            float t8 = time_ms([&] { k_mfma_asm<8><<<nblk, 256>>>(out, it2, 1e-3); });
            const double fl = (double)nblk * 4 * it2 * 2048.0;
            printf("MFMA f64 (asm, VGPR acc) %d wave(s)/SIMD: 1 acc %.1f TF | 2 acc %.1f TF | 4 acc %.1f TF | 8 acc %.1f TF\n", wps,
                   fl * 1 / t1 / 1e9, fl * 2 / t2 / 1e9, fl * 4 / t4 / 1e9, fl * 8 / t8 / 1e9);
        }
    }
    return 0;
}
