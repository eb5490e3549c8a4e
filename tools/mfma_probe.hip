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

// The statistics-kernel inner loop in isolation: 16 accumulators (4 x [S0 S1 S2_0 S2_1]), per "row" r
// 4 B operands, each feeding 4 MFMAs.  MODE 0: B operands stay in registers; 1: every B operand is a
// fresh ds_read_b64; 2: additionally b2 = b * b on the VALU; 3: additionally the A operand is a
// fresh VALU product; 4: additionally the A operands stream from global memory (2 x 32 B per lane and
// 16-frame block, prefetched one block ahead, like k_stats_z); 5: additionally a __syncthreads() and a
// register-staged LDS tile write per 4 blocks.  One 8-wave workgroup per CU like k_stats_z.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_lds_mfma(double *out, int iters, double seed, const double *stream)
{
    __shared__ double tile[2][64 * 96];
    for (int i = threadIdx.x; i < 2 * 64 * 96; i += 512) tile[0][i] = seed * (1 + (i & 7));
    __syncthreads();
    const int lane = threadIdx.x & 63, i16 = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    d4 S[2][4], S2[2][4];
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 4; ++j) { S[t][j] = (d4){seed, seed, seed, seed}; S2[t][j] = S[t][j]; }
    double g0 = seed + lane * 1e-9, g1 = seed * 0.5, f = 1.0 + seed;
    double breg[4] = {seed, seed * 2, seed * 3, seed * 4};
    // wave-private stream: iters * 4 blocks of 2 x 2 KB
    const double *zp = stream + ((size_t)(blockIdx.x * 8 + wave) * iters * 4 * 2 * 64 + lane) * 4;
    d4 zn0 = (d4){seed, seed, seed, seed}, zn1 = zn0;
    if (MODE >= 4) { zn0 = __builtin_nontemporal_load((const d4 *)zp); zn1 = __builtin_nontemporal_load((const d4 *)(zp + 256)); }
    float stg[8];
    for (int it = 0; it < iters; ++it) {
        const double *pS = tile[it & 1] + q * 96 + ((q & 1) << 4) + ((q >> 1) << 1) + i16;
        if (MODE >= 5)
#pragma unroll
            for (int i = 0; i < 8; ++i) stg[i] = ((const float *)stream)[(size_t)(blockIdx.x * iters + it) * 4096 + threadIdx.x + 512 * i];
#pragma unroll
        for (int fs = 0; fs < 4; ++fs) {
            d4 zc0 = zn0, zc1 = zn1;
            if (MODE >= 4) {
                const size_t n = (size_t)it * 4 + fs + 1;
                if (n < (size_t)iters * 4) {
                    zn0 = __builtin_nontemporal_load((const d4 *)(zp + n * 512));
                    zn1 = __builtin_nontemporal_load((const d4 *)(zp + n * 512 + 256));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double a0 = g0, a1 = g1;
                if (MODE == 3) { a0 = g0 * f; a1 = g1 * f; g0 = a1; g1 = a0; }
                if (MODE >= 4) { a0 = zc0[r] * f; a1 = zc1[r] * f; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double bv = MODE >= 1 ? pS[(fs * 16 + 4 * r) * 96 + 4 * r + 16 * j] : breg[j];
                    S[0][j] = MFMA_F64(a0, bv, S[0][j]);
                    S[1][j] = MFMA_F64(a1, bv, S[1][j]);
                    const double b2 = MODE >= 2 ? bv * bv : bv;
                    S2[0][j] = MFMA_F64(a0, b2, S2[0][j]);
                    S2[1][j] = MFMA_F64(a1, b2, S2[1][j]);
                }
            }
        }
        if (MODE >= 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) tile[(it + 1) & 1][(threadIdx.x + 512 * i) % (64 * 96)] = (double)stg[i];
            __syncthreads();
        } else if (MODE >= 1) asm volatile("" ::: "memory"); // keep the LDS reads inside the loop
    }
    double s = g0 + g1;
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 4; ++j) s += S[t][j][0] + S[t][j][1] + S[t][j][2] + S[t][j][3] + S2[t][j][0] + S2[t][j][1] + S2[t][j][2] + S2[t][j][3];
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
    {
        const int it2 = 400, nblk = p.multiProcessorCount;
        auto tf = [&](float ms) { return (double)nblk * 8 * it2 * 256.0 * 2048.0 / ms / 1e9; };
        double *stream = nullptr;
        const size_t sbytes = (size_t)nblk * 8 * it2 * 4 * 4096 + (1 << 20);
        if (hipMalloc(&stream, sbytes) != hipSuccess) { printf("stream alloc failed\n"); return 1; }
        hipMemset(stream, 0, sbytes);
        float t0 = time_ms([&] { k_lds_mfma<0><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        float t1 = time_ms([&] { k_lds_mfma<1><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        float t2 = time_ms([&] { k_lds_mfma<2><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        float t3 = time_ms([&] { k_lds_mfma<3><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        float t4 = time_ms([&] { k_lds_mfma<4><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        float t5 = time_ms([&] { k_lds_mfma<5><<<nblk, 512>>>(out, it2, 1e-3, stream); });
        printf("statistics inner loop (8 waves/CU, 16 accumulators): regs only %.1f TF | + ds_read B %.1f TF | + b*b %.1f TF | + A product %.1f TF"
               " | + A streamed from HBM (%.1f GB) %.1f TF | + staged tile and barrier per 4 blocks %.1f TF\n",
               tf(t0), tf(t1), tf(t2), tf(t3), sbytes / 1e9, tf(t4), tf(t5));
        hipFree(stream);
    }
    return 0;
}
