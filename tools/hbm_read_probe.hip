// hbm_read_probe.hip -- streaming-read bandwidth of one MI355X as a function of the footprint and the load flavour (plain / non-temporal),
// 16 bytes per lane, every workgroup walks its own contiguous slab (like a k_stats_z wave walks its likelihood tile).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_probe.hip -o tools/bin/hbm_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT, bool SLAB>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ p, size_t n16, double *out)
{
    d2 s = {0, 0};
    if (SLAB) { // contiguous slab per workgroup
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t b = per * blockIdx.x, e = b + per < n16 ? b + per : n16;
        for (size_t i = b + threadIdx.x; i < e; i += 256) { const d2 v = NT ? __builtin_nontemporal_load(p + i) : p[i]; s += v; }
    } else {    // grid-stride
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const d2 v = NT ? __builtin_nontemporal_load(p + i) : p[i]; s += v; }
    }
    if (s[0] + s[1] == 12345.678) out[0] = s[0];
}
// the arithmetic of k_frame_moments on the same stream: float4 vectors, 4 conversions + 4 sums + 4 sums of squares in fp64 per vector
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> // 0: sums of the raw floats (fp32 adds), 1: fp64 conversions + sums, 2: + sums of squares
__global__ __launch_bounds__(256) void k_moments_like(const f4 *__restrict__ p, size_t n16, double *out)
{
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    f4 sf = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (MODE != 3) for (; i + 3 * stride < n16; i += 4 * stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) sf += v[u];
            else
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double d = (double)v[u][k]; s[k] += d; if (MODE == 2) ss[k] = __builtin_fma(d, d, ss[k]); }
        }
    }
    if (MODE == 3) { // K4's arithmetic, ONE load per trip (the compiler unrolls and pipelines as it sees fit)
        for (i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            const f4 v = __builtin_nontemporal_load(p + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double d = (double)v[k]; s[k] += d; ss[k] = __builtin_fma(d, d, ss[k]); }
        }
    }
    const double t = s[0] + s[1] + s[2] + s[3] + ss[0] + ss[1] + ss[2] + ss[3] + sf[0] + sf[1] + sf[2] + sf[3];
    if (t == 12345.678) out[0] = t;
}
template <int MODE> static void run_m(const char *name, const d2 *p, size_t bytes, double *out, int grid)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_moments_like<MODE><<<grid, 256>>>((const f4 *)p, bytes / 16, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k_moments_like<MODE><<<grid, 256>>>((const f4 *)p, bytes / 16, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %6.1f GB  grid %6d  %7.3f ms  %5.2f TB/s\n", name, bytes / 1e9, grid, ms / 3, bytes / (ms / 3) / 1e9);
}
template <bool NT, bool SLAB> static void run(const char *name, const d2 *p, size_t bytes, double *out, int grid)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_read<NT, SLAB><<<grid, 256>>>(p, bytes / 16, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k_read<NT, SLAB><<<grid, 256>>>(p, bytes / 16, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %6.1f GB  grid %6d  %7.3f ms  %5.2f TB/s\n", name, bytes / 1e9, grid, ms / 3, bytes / (ms / 3) / 1e9);
}
int main()
{
    double *out;
    (void)hipMalloc(&out, 64);
    for (size_t gb : {2, 8, 32, 64}) {
        const size_t bytes = gb << 30;
        d2 *p;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %zu GB failed\n", gb); continue; }
        (void)hipMemset(p, 0, bytes);
        if (gb == 2) {
            run_m<0>("float4 sums (fp32)", p, bytes, out, 2040);
            run_m<1>("fp64 conversions + sums", p, bytes, out, 2040);
            run_m<2>("+ sums of squares (K4's arithmetic)", p, bytes, out, 2040);
            run_m<3>("K4's arithmetic, one load per trip", p, bytes, out, 2040);
            run_m<3>("K4's arithmetic, one load per trip", p, bytes, out, 8160);
        }
        for (int grid : {2048, 16384}) {
            run<false, false>("plain, grid-stride", p, bytes, out, grid);
            run<true, false>("non-temporal, grid-stride", p, bytes, out, grid);
            run<false, true>("plain, slab per workgroup", p, bytes, out, grid);
            run<true, true>("non-temporal, slab per workgroup", p, bytes, out, grid);
        }
        (void)hipFree(p);
    }
    return 0;
}
