// valu_rate_probe.hip -- issue cost (cycles per wave64 instruction and SIMD) of the fp64 VALU instructions of K1's exponential:
// one wave per SIMD, 8 independent chains, 4096 x 8 instructions each.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 4096
template <int OP> __global__ __launch_bounds__(256) void k(double *out, double a, double b, int ia)
{
    double v[8];
    int iv[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + threadIdx.x * 1e-3 + i; iv[i] = ia + i; }
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[i]) : "v"(v[i]));
            if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(iv[i]));
            if (OP == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(iv[i]) : "v"(ia));
            if (OP == 9) asm volatile("v_ashrrev_i32 %0, 11, %0" : "+v"(iv[i]));
            if (OP == 10) asm volatile("v_max_i32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
            if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[i]) : "v"(ia));
            if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(iv[i]) : "v"(ia) : "s20", "s21");
            if (OP == 16) asm volatile("v_cmp_lt_i32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[i]) : "v"(ia) : "vcc");
            if (OP == 17) asm volatile("v_cmp_lt_i32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(iv[i]) : "v"(ia) : "s20", "s21");
            if (OP == 18) asm volatile("v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(iv[i]));
            if (OP == 19) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
            if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[i]) : "v"(ia));
            if (OP == 12) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[i]) : "v"(iv[i]));
            if (OP == 13) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc");
            if (OP == 14) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(iv[i]) : "v"(v[i]));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4; // 4 workgroups of 4 waves per CU: one wave per SIMD x 4 rounds
    k<OP><<<grid, 256>>>(d, 1.0000001, 0.9999999, 3);
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(d, 1.0000001, 0.9999999, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 rounds... each CU runs 4 workgroups = 16 waves = 4 per SIMD, each wave REP * 8 instructions
    const double inst_per_simd = 4.0 * REP * 8;
    printf("%-16s %.3f ms  -> %.2f cycles per instruction and SIMD at 2.4 GHz (4 waves per SIMD interleaved)\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
}
int main()
{
    double *d; hipMalloc(&d, 256 * 1024 * 8);
    run<0>("v_fma_f64", d); run<1>("v_mul_f64", d); run<2>("v_add_f64", d); run<3>("v_rndne_f64", d); run<4>("v_cvt_i32_f64", d);
    run<5>("v_ldexp_f64", d); run<6>("v_max_f64", d); run<12>("v_cvt_f64_i32", d); run<13>("v_cmp_ge_f64", d); run<14>("v_cvt_f32_f64", d);
    run<7>("v_and_b32", d); run<8>("v_lshl_add_u32", d); run<9>("v_ashrrev_i32", d); run<10>("v_max_i32", d); run<11>("v_cndmask_b32 vcc", d);
    run<15>("v_cndmask_e64 sgpr", d); run<16>("cmp+cndmask vcc (2)", d); run<17>("cmp+cndmask sgpr (2)", d); run<18>("v_max_i32_dpp", d);
    run<19>("v_sub_u32", d); run<20>("v_mov_b32", d);
    return 0;
}
