// tv_kernels.hip -- fp64 building blocks of the i-vector path on gfx950: a strided-batched MFMA
// GEMM (v_mfma_f64_16x16x4_f64, LDS-tiled 128x128x16), a batched blocked Cholesky / SPD inverse
// built from it, and the small element-wise / packing kernels around them.
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#include "devutil.h"
#include "lds_attr.h"
#include "tv_kernels.h"

// -------------------------------------------------------------------------------------------
// C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]       (row-major, element strides)
//   op(A)[m][k] = TA ? A[k*lda + m] : A[m*lda + k]     op(B)[k][n] = TB ? B[n*ldb + k] : B[k*ldb + n]
// Workgroup = 4 waves, tile 128 x 128, each wave 64 x 64 (4 x 4 MFMA tiles), BK = 16.
// Both operand tiles live k-major in LDS, column index XOR-swizzled by
//   f(k) = (k & 15) ^ ((k & 1) << 4)
// which keeps the MFMA operand reads (16 consecutive columns x 2 adjacent k) and both staging
// write patterns (along columns, or transposing along k) free of bank conflicts without padding.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ int kswz(int k) { return (k & 15) ^ ((k & 1) << 4); }

// Two instantiations per transpose pair:
//   EDGE = false  interior tiles (full 128 x 128, K range a multiple of 16, rows 16-byte aligned): no bounds
//                 checks, 16-byte loads, one base pointer per operand advanced per k-tile.  The register
//                 budget is 256 unified VGPRs (2 waves per SIMD): with 512 the compiler parks the
//                 accumulators in AGPRs and pays 2 v_accvgpr moves per MFMA (measured: core loop 56 instead of
//                 73 TFLOP/s, tools/gemm_probe.hip).
//   EDGE = true   tiles cut by M, N or K: per-element loads with bounds checks (the original path).
// m_off / n_off: element offsets of the launched sub-grid (interior, right strip, bottom strip).
// Fused epilogue on the accumulator v = alpha * acc (before the beta term):
//   mode 1: v * rv[m] * cv[n]                         (cosine scoring: the two reciprocal norms)
//   mode 2: v + br * rv[m] + bc * cv[n] + cst         (quadratic-form scoring rules: the per-model / per-segment terms)
// One pass over the M x N result instead of GEMM + a read-modify-write kernel (3.2 GB each way at 20 k x 20 k).
struct DgemmEpi { const double *rv, *cv; double br, bc, cst; int mode; int remap; };

// AM x AN: 16 x 16 MFMA tiles per wave (2 x 2 waves): the workgroup tile is 32 AM x 32 AN.  4 x 4 everywhere except on the strips that
// M or N cut off a 128-multiple (MODE 2 only): 1 x 4 (32 x 128) below, 4 x 1 (128 x 32) to the right -- with R = 400 = 3 x 128 + 16
// a 128-row strip tile spends 7/8 of its MFMAs on clamped duplicates (Cmx += W^T F at 400 x 122880 x 1024: 2.04 ms against 1.55 ms
// for 384 rows).
// WM x WN: the wave grid of the workgroup (2 x 2 everywhere but one shape).  <AM 2, AN 5, WM 4, WN 1> is a 128 x 80 tile for NT products
// whose N is a multiple of 80 but not of 128 -- R = 400 = 5 x 80: `aux = F (T Sigma^-1)^T` (1024 x 400 x 122880, split-K) ran three
// 128-wide tile columns plus a 16-column strip that re-read all of F beside the interior grid (2.0 ms where the interior alone takes
// 1.5); with 80-wide tiles there is no strip.  The B tile is staged 96 rows wide (three passes of 32 rows, the last one clamped to
// the tile's own last row) because the XOR swizzle moves a column inside its group of 32.
template <bool TA, bool TB, int MODE, int AM = 4, int AN = 4, int WM = 2, int WN = 2>
__global__ __launch_bounds__(256, (MODE == 1 ? 1 : 2)) void k_dgemm(int M, int N, int K, double alpha, const double *__restrict__ A,
                                                  long lda, long sA, const double *__restrict__ B, long ldb, long sB,
                                                  double beta, double *__restrict__ C, long ldc, long sC, int ksplit, int m_off, int n_off,
                                                  DgemmEpi epi)
{
    constexpr int BM = 16 * AM * WM, BN = 16 * AN * WN, BK = 16;
    constexpr int BNP = (BN + 31) / 32 * 32;              // staged width of the B tile (whole passes of 32 rows / columns)
    static_assert(WM * WN == 4, "four waves");
    static_assert(MODE == 2 || (AM == 4 && AN == 4) || (WM == 4 && WN == 1), "narrow tiles exist for the clamped strips only");
    static_assert((WM == 2 && WN == 2) || (MODE == 0 && !TA && TB && BM % 32 == 0), "the 4 x 1 wave grid serves interior NT tiles");
    constexpr int NLA = BM / 32, NLB = BNP / 32;          // 16-byte staging loads per thread and k-tile
    constexpr int HA = BM / 2, HB = BN / 2;               // column pairs of an m- (n-) fastest operand tile
    constexpr int KSA = 256 / HA, KSB = 256 / HB;         // k rows covered by one such load of the workgroup
    __shared__ __attribute__((aligned(16))) double As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) double Bs[2][BK][BNP];
    typedef double d2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;
    // Tile order.  Hardware order is blockIdx.x (N tiles) fastest and workgroup i lands on XCD i % 8: the ~512 resident tiles
    // of a wide GEMM then share one row panel of op(A) and stream 512 different column panels of op(B) -- every B panel comes
    // from HBM once per M tile row.  remap: XCD x takes the N tiles n = 8 g + x and walks the M tiles fastest, so a B panel is
    // fetched once (by one XCD's L2) and reused for the whole column of tiles; the A panels cycle through L2 / MALL.
    // remap 2 (round 5, A/B knob): the same split of the N tiles over the XCDs, but an XCD walks its (M tile, own N tile) plane in
    // BLOCKS of 8 x 8 tiles -- its ~64 resident workgroups then share 8 A panels and 8 B panels instead of 64 A panels and one B
    // panel.  Tried because the 100 k x 100 k scoring GEMM (782 x 782 tiles of K = 400) fetches 127 M KB (FETCH_SIZE) per call against
    // 0.64 GB of operands; measured: FETCH_SIZE 123.7 M KB against 124.8 M, 126.2 ms against 126.3 -- the L2 captures the same quarter
    // of the 501 GB the tiles ask for in either order, and the kernel is not bound by it (0.79 of the MFMA peak, 2 TB/s of fetches +
    // 0.63 TB/s of score writes).  Not the default.
    int bx = blockIdx.x, by = blockIdx.y;
    if (epi.remap) {
        const int Nt = gridDim.x, Mt = gridDim.y, G = Nt >> 3;
        const int id = by * Nt + bx;
        if (id < G * 8 * Mt && epi.remap >= 2) {
            const int xcd = id & 7, loc = id >> 3;   // loc: this XCD's own sequence over Mt x G tiles
            const int bn = loc / (8 * Mt);           // block column (8 of the XCD's N tiles; the last one may be narrower)
            const int w = G - 8 * bn < 8 ? G - 8 * bn : 8;
            const int r = loc - bn * 8 * Mt;
            const int bm = r / (8 * w);              // block row inside the column (the last one may be shorter)
            const int h = Mt - 8 * bm < 8 ? Mt - 8 * bm : 8;
            const int rr = r - bm * 8 * w;
            by = 8 * bm + rr % h;
            bx = (8 * bn + rr / h) * 8 + xcd;
        } else if (id < G * 8 * Mt) {
            const int xcd = id & 7, loc = id >> 3;
            by = loc % Mt;
            bx = (loc / Mt) * 8 + xcd;
        } else {
            const int r = id - G * 8 * Mt;
            by = r % Mt;
            bx = G * 8 + r / Mt;
        }
    }
    const long m0 = (long)by * BM + m_off, n0 = (long)bx * BN + n_off;
    // ksplit > 0: blockIdx.z selects a K range [kb, ke) and C is the z-th partial slab (stride sC);
    // otherwise blockIdx.z is the batch index.
    long kb = 0, ke = K;
    if (ksplit > 0) {
        kb = (long)blockIdx.z * ksplit;
        ke = kb + ksplit < K ? kb + ksplit : K;
    } else {
        A += (size_t)blockIdx.z * sA;
        B += (size_t)blockIdx.z * sB;
    }
    C += (size_t)blockIdx.z * sC;

    // ---- staging: EDGE -> 8 + 8 checked scalar loads; interior -> 4 + 4 unchecked 16-byte loads -----------
    constexpr bool EDGE = MODE == 1; // per-element checked loads
    constexpr bool CHECK_OUT = MODE != 0;
    double ra[EDGE ? 8 : 2 * NLA], rb[EDGE ? 8 : 2 * NLB];
    // interior path: operand with k fastest in memory: thread owns k pair kq = 2 (tid & 7), rows (tid >> 3) + 32 i;
    //                operand with m (n) fastest:       thread owns column pair 2 (tid % HA), k rows tid / HA + KSA i  (128 wide: 2 (tid & 63), (tid >> 6) + 4 i)
    // MODE 2 (tiles cut by M or N, K range still a multiple of 16): the same loads with the row / column-pair index CLAMPED
    // into the matrix -- an output element depends only on its own row of op(A) and column of op(B), so the clamped
    // duplicates only feed outputs that are not stored.  No check inside the k loop.
    const double *pa = nullptr, *pb = nullptr;
    long oa[4] = {0, 0, 0, 0}, ob[4] = {0, 0, 0, 0}; // element offsets of the (up to 4) loads of a k-tile
    if (!EDGE) {
        long ma = m0 + (TA ? 2 * (tid % HA) : (tid >> 3)), nb_ = n0 + (TB ? (tid >> 3) : 2 * (tid % HB));
        if (MODE == 2) {
            if (TA) ma = ma < M - 2 ? ma : M - 2;
            if (!TB) nb_ = nb_ < N - 2 ? nb_ : N - 2;
        }
        pa = TA ? A + (kb + tid / HA) * lda + ma : A + kb + 2 * (tid & 7);
        pb = TB ? B + kb + 2 * (tid & 7) : B + (kb + tid / HB) * ldb + nb_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long ri = ma + 32 * i, ci = nb_ + 32 * i;
            if (MODE == 2) { ri = ri < M - 1 ? ri : M - 1; ci = ci < N - 1 ? ci : N - 1; }
            if (BNP != BN) ci = ci < n0 + BN - 1 ? ci : n0 + BN - 1; // the padded part of the last staging pass: the tile's own last row again
            oa[i] = TA ? (long)(KSA * i) * lda : ri * lda;
            ob[i] = TB ? ci * ldb : (long)(KSB * i) * ldb;
        }
    }
    auto gload = [&](int kt) {
        if (!EDGE) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                const d2 va = *(const d2 *)(pa + oa[i]);
                ra[2 * i] = va[0]; ra[2 * i + 1] = va[1];
            }
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                const d2 vb = *(const d2 *)(pb + ob[i]);
                rb[2 * i] = vb[0]; rb[2 * i + 1] = vb[1];
            }
            pa += TA ? (long)BK * lda : BK;
            pb += TB ? BK : (long)BK * ldb;
            return;
        }
        const long k0 = kb + (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int k, m;
            if (TA) { m = e & 127; k = e >> 7; } else { k = e & 15; m = e >> 4; }
            const long gm = m0 + m, gk = k0 + k;
            double v = 0.0;
            if (gm < M && gk < ke) v = TA ? A[gk * lda + gm] : A[gm * lda + gk];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int k, n;
            if (TB) { k = e & 15; n = e >> 4; } else { n = e & 127; k = e >> 7; }
            const long gn = n0 + n, gk = k0 + k;
            double v = 0.0;
            if (gn < N && gk < ke) v = TB ? B[gn * ldb + gk] : B[gk * ldb + gn];
            rb[i] = v;
        }
    };
    // the last k-tile when the K range is not a multiple of 16 (K even): pairs / rows beyond the range are fetched from a
    // valid element of the tile and count as zero -- estimateTETt (K = D = 60) and every other odd-sized product stay on
    // the 16-byte-load instantiations instead of the per-element checked one
    const int krem = EDGE ? 0 : (int)((ke - kb) & 15);
    auto gload_tail = [&]() {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int ka = TA ? tid / HA + KSA * i : 2 * (tid & 7);
            const bool oka = ka < krem;
            const double *qa = pa + oa[i] - (oka ? 0L : (TA ? (long)ka * lda : (long)ka));
            const d2 va = *(const d2 *)qa;
            ra[2 * i] = oka ? va[0] : 0.0; ra[2 * i + 1] = oka ? va[1] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int kb_ = TB ? 2 * (tid & 7) : tid / HB + KSB * i;
            const bool okb = kb_ < krem;
            const double *qb = pb + ob[i] - (okb ? 0L : (TB ? (long)kb_ : (long)kb_ * ldb));
            const d2 vb = *(const d2 *)qb;
            rb[2 * i] = okb ? vb[0] : 0.0; rb[2 * i + 1] = okb ? vb[1] : 0.0;
        }
    };
    auto swrite = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        if (!EDGE) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                if (TA) { const int k = tid / HA + KSA * i, m = 2 * (tid % HA); As[buf][k][m ^ kswz(k)] = ra[2 * i]; As[buf][k][(m + 1) ^ kswz(k)] = ra[2 * i + 1]; }
                else { const int k = 2 * (tid & 7), m = (tid >> 3) + 32 * i; As[buf][k][m ^ kswz(k)] = ra[2 * i]; As[buf][k + 1][m ^ kswz(k + 1)] = ra[2 * i + 1]; }
            }
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                if (TB) { const int k = 2 * (tid & 7), n = (tid >> 3) + 32 * i; Bs[buf][k][n ^ kswz(k)] = rb[2 * i]; Bs[buf][k + 1][n ^ kswz(k + 1)] = rb[2 * i + 1]; }
                else { const int k = tid / HB + KSB * i, n = 2 * (tid % HB); Bs[buf][k][n ^ kswz(k)] = rb[2 * i]; Bs[buf][k][(n + 1) ^ kswz(k)] = rb[2 * i + 1]; }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int k, m;
            if (TA) { m = e & 127; k = e >> 7; } else { k = e & 15; m = e >> 4; }
            As[buf][k][m ^ kswz(k)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int k, n;
            if (TB) { k = e & 15; n = e >> 4; } else { n = e & 127; k = e >> 7; }
            Bs[buf][k][n ^ kswz(k)] = rb[i];
        }
    };

    d4 acc[AM][AN];
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < AN; ++b) acc[a][b] = (d4){0, 0, 0, 0};

    const int nkt = (int)((ke - kb + BK - 1) / BK);
    if (nkt > 0) {
        if (krem && nkt == 1) gload_tail();
        else gload(0);
        swrite(std::integral_constant<int, 0>{});
    }
    __syncthreads();
    // The k-tile loop is unrolled by two so that the LDS buffer index is a compile-time constant: the 32 operand reads and 16
    // staging writes of a k-tile then address "lane-dependent register + immediate" -- with a run-time buffer index every one
    // of them paid a v_lshl_add to rebuild its address (48 VALU instructions next to 64 MFMAs).
    auto ktile = [&](auto curc, int kt) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value;
        if (kt + 1 < nkt) {
            if (krem && kt + 2 == nkt) gload_tail();
            else gload(kt + 1);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + q;
            const int sw = kswz(k);
            double av[AM], bv[AN];
#pragma unroll
            for (int a = 0; a < AM; ++a) av[a] = As[cur][k][(wr * (16 * AM) + a * 16 + i16) ^ sw];
#pragma unroll
            for (int b = 0; b < AN; ++b) bv[b] = Bs[cur][k][(wc * (16 * AN) + b * 16 + i16) ^ sw];
#pragma unroll
            for (int a = 0; a < AM; ++a)
#pragma unroll
                for (int b = 0; b < AN; ++b) acc[a][b] = MFMA_F64(av[a], bv[b], acc[a][b]);
        }
        if (kt + 1 < nkt) swrite(std::integral_constant<int, cur ^ 1>{});
        __syncthreads();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nkt) ktile(std::integral_constant<int, 1>{}, kt + 1);
    }
    // D layout: lane holds rows q + 4r, column i16 of every 16x16 tile
    double cvv[AN];
#pragma unroll
    for (int b = 0; b < AN; ++b) cvv[b] = 0.0;
    if (epi.mode != 0) {
#pragma unroll
        for (int b = 0; b < AN; ++b) {
            const long gn = n0 + wc * (16 * AN) + b * 16 + i16;
            cvv[b] = (!CHECK_OUT || gn < N) ? epi.cv[gn] : 0.0;
        }
    }
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        // beta != 0: the 16 old values of this row block are requested TOGETHER, before the first store -- read one by one inside
        // the store loop every load waits behind the previous store to the same array (may alias: the compiler keeps the order
        // and each load pays its full latency; A += N^T E at K = 1024: 5.64 -> 5.13 ms)
        double cin[4][AN];
        if (beta != 0.0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long gm = m0 + wr * (16 * AM) + a * 16 + q + 4 * r;
#pragma unroll
                for (int b = 0; b < AN; ++b) {
                    const long gn = n0 + wc * (16 * AN) + b * 16 + i16;
                    cin[r][b] = (!CHECK_OUT || (gm < M && gn < N)) ? C[gm * ldc + gn] : 0.0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long gm = m0 + wr * (16 * AM) + a * 16 + q + 4 * r;
            if (CHECK_OUT && gm >= M) continue;
            const double rvv = epi.mode != 0 ? epi.rv[gm] : 0.0;
#pragma unroll
            for (int b = 0; b < AN; ++b) {
                const long gn = n0 + wc * (16 * AN) + b * 16 + i16;
                if (CHECK_OUT && gn >= N) continue;
                double v = alpha * acc[a][b][r];
                if (epi.mode == 1) v = v * rvv * cvv[b];
                else if (epi.mode == 2) v = v + epi.br * rvv + epi.bc * cvv[b] + epi.cst;
                if (beta != 0.0) C[gm * ldc + gn] = v + beta * cin[r][b];
                else if (epi.mode != 0) __builtin_nontemporal_store(v, &C[gm * ldc + gn]); // score matrices (GBs, written once): streamed past L2
                else C[gm * ldc + gn] = v;
            }
        }
    }
}

// m_off / n_off: element offsets of the launched sub-grid (interior, right strip, bottom strip)
template <int MODE, int AM = 4, int AN = 4>
static void launch_dgemm_e(hipStream_t st, bool ta, bool tb, dim3 grid, int M, int N, int K, double alpha, const double *A,
                           long lda, long sA, const double *B, long ldb, long sB, double beta, double *C, long ldc, long sC,
                           int ksplit, int m_off, int n_off, const DgemmEpi &epi)
{
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
    if (!ta && !tb) k_dgemm<false, false, MODE, AM, AN><<<grid, 256, 0, st>>>(M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, m_off, n_off, epi);
    else if (!ta && tb) k_dgemm<false, true, MODE, AM, AN><<<grid, 256, 0, st>>>(M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, m_off, n_off, epi);
    else if (ta && !tb) k_dgemm<true, false, MODE, AM, AN><<<grid, 256, 0, st>>>(M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, m_off, n_off, epi);
    else k_dgemm<true, true, MODE, AM, AN><<<grid, 256, 0, st>>>(M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, m_off, n_off, epi);
}

// NT product on 128 x 80 tiles (no bounds checks at all): M a multiple of 128, N of 80, every K range of 16, 16-byte aligned rows
static void launch_dgemm_nt80(hipStream_t st, int M, int N, int K, double alpha, const double *A, long lda, const double *B, long ldb,
                              double beta, double *C, long ldc, long sC, int ksplit, int nz)
{
    DgemmEpi epi{nullptr, nullptr, 0.0, 0.0, 0.0, 0, 0};
    k_dgemm<false, true, 0, 2, 5, 4, 1><<<dim3(N / 80, M / 128, nz), 256, 0, st>>>(M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, sC, ksplit, 0, 0, epi);
}

// A/B knobs of the calling context (ctx.h: gmmiv_kopts, bound per call): XCD-aware tile order; 0 = cut tiles always on the
// per-element checked instantiation; 0 = 128 x 128 tiles on the strips too
#define g_gemm_remap (gmmiv_kopts_cur().gemm_remap)
#define g_gemm_clamp (gmmiv_kopts_cur().gemm_clamp)
#define g_gemm_narrow (gmmiv_kopts_cur().gemm_narrow)

// grid.z = batch (ksplit == 0) or K layers (ksplit > 0).  Three instantiations: full tiles run MODE 0 (no checks at all) when
// every K range is a multiple of 16 and the operands allow 16-byte loads; tiles cut by M or N run MODE 2 (the same loads,
// indices clamped, checked stores) when the m- / n-fastest operands have an even extent; everything else MODE 1 (per-element
// checks).
static void launch_dgemm(hipStream_t st, bool ta, bool tb, dim3 grid, int M, int N, int K, double alpha, const double *A,
                         long lda, long sA, const double *B, long ldb, long sB, double beta, double *C, long ldc, long sC,
                         int ksplit, DgemmEpi epi = DgemmEpi{nullptr, nullptr, 0.0, 0.0, 0.0, 0, 0})
{
    epi.remap = g_gemm_remap && grid.x >= 16; // wide enough for the per-XCD column order to mean something
    // a partial last k-tile is handled in the kernel; K must be even only when an operand has k as its FASTEST index (16-byte pairs
    // along k).  A^T B products (ta && !tb: `A += N^T E`, `Cmx += W^T F` with K = the number of utterances) take any K -- with an odd
    // utterance count they used to fall to the per-element checked instantiation as a whole
    const bool kfull = (K % 2 == 0 || (ta && !tb)) && (ksplit <= 0 || ksplit % 16 == 0);
    const bool aligned = (((size_t)A | (size_t)B) % 16 == 0) && lda % 2 == 0 && ldb % 2 == 0 && sA % 2 == 0 && sB % 2 == 0;
    const bool clamp_ok = g_gemm_clamp && (!ta || (M % 2 == 0 && M >= 2)) && (tb || (N % 2 == 0 && N >= 2));
    const int fm = M / 128, fn = N / 128; // full tiles
    if (!kfull || !aligned || K <= 0) {
        launch_dgemm_e<1>(st, ta, tb, grid, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, 0, epi);
        return;
    }
    if (fm == 0 || fn == 0) { // no full tile at all
        // a handful of rows (one utterance's L = N TETt: 1 x 80200 x 2048) on 32-row tiles: a 128-row tile would spend 127 / 128 of its MFMAs
        // on clamped duplicates (0.97 ms where the 1.3 GB of TETt stream in 0.3)
        if (clamp_ok && g_gemm_narrow && M <= 64 && N > 128)
            launch_dgemm_e<2, 1, 4>(st, ta, tb, dim3(grid.x, (M + 31) / 32, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, 0, epi);
        else if (clamp_ok) launch_dgemm_e<2>(st, ta, tb, grid, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, 0, epi);
        else launch_dgemm_e<1>(st, ta, tb, grid, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, 0, epi);
        return;
    }
    // The strips are few workgroups that each walk the whole K range: on the caller's stream they would run AFTER the
    // interior grid and add their full latency.  They go to a side stream (forked and joined with events) and run
    // beside the interior tiles; the tiles are disjoint, so there is no ordering between the launches to keep.
    const bool has_strips = (int)grid.x > fn || (int)grid.y > fm;
    // one side stream + event pair per device AND per host thread (a process may hold contexts on several devices, and two contexts
    // on one device may be driven from two threads at once: a shared event pair would let one thread's fork overwrite the other's)
    struct Side { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
    static thread_local Side sides[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool forked = false;
    if (has_strips && dev >= 0) {
        Side &sd = sides[dev];
        if (!sd.s) {
            if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess) sd.s = nullptr;
            else if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(sd.s); sd.s = nullptr; }
        }
        side = sd.s; ev_fork = sd.fork; ev_join = sd.join;
        forked = side && hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess;
    }
    hipStream_t ss = forked ? side : st;
    if (clamp_ok) {
        // strips of at most 64 rows / columns run 32-wide tiles (g_gemm_narrow: A/B knob)
        const int rm = M - fm * 128, rn = N - fn * 128;
        if ((int)grid.x > fn) {
            if (g_gemm_narrow && rn <= 64) launch_dgemm_e<2, 4, 1>(ss, ta, tb, dim3((rn + 31) / 32, grid.y, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, fn * 128, epi);
            else launch_dgemm_e<2>(ss, ta, tb, dim3(grid.x - fn, grid.y, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, fn * 128, epi);
        }
        if ((int)grid.y > fm) {
            if (g_gemm_narrow && rm <= 64) launch_dgemm_e<2, 1, 4>(ss, ta, tb, dim3(fn, (rm + 31) / 32, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, fm * 128, 0, epi);
            else launch_dgemm_e<2>(ss, ta, tb, dim3(fn, grid.y - fm, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, fm * 128, 0, epi);
        }
    } else {
        if ((int)grid.x > fn) launch_dgemm_e<1>(ss, ta, tb, dim3(grid.x - fn, grid.y, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, fn * 128, epi);
        if ((int)grid.y > fm) launch_dgemm_e<1>(ss, ta, tb, dim3(fn, grid.y - fm, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, fm * 128, 0, epi);
    }
    launch_dgemm_e<0>(st, ta, tb, dim3(fn, fm, grid.z), M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, ksplit, 0, 0, epi);
    if (forked) {
        (void)hipEventRecord(ev_join, side);
        (void)hipStreamWaitEvent(st, ev_join, 0);
    }
}

int tvk_dgemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda, long sA,
              const double *B, long ldb, long sB, double beta, double *C, long ldc, long sC, int batch)
{
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    dim3 grid((N + 127) / 128, (M + 127) / 128, batch);
    launch_dgemm(st, ta, tb, grid, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, 0);
    return (int)hipGetLastError();
}
// C = epilogue(alpha op(A) op(B)): mode 1 -> x rv[m] cv[n]; mode 2 -> + br rv[m] + bc cv[n] + cst   (single matrix)
int tvk_dgemm_epi(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda, const double *B,
                  long ldb, double *C, long ldc, int mode, const double *rv, const double *cv, double br, double bc, double cst, double beta)
{
    if (M <= 0 || N <= 0) return 0;
    dim3 grid((N + 127) / 128, (M + 127) / 128, 1);
    launch_dgemm(st, ta, tb, grid, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 0, DgemmEpi{rv, cv, br, bc, cst, mode, 0});
    return (int)hipGetLastError();
}

// scores[m][s] = fill wherever trials[m][s] == 0 (PldaTest::_trials: only the listed trials are scored, PldaTools.cpp:3871, 3889)
__global__ void k_mask_trials(long n, const unsigned char *__restrict__ trials, double fill, double *__restrict__ scores)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        if (!trials[e]) scores[e] = fill;
}
int tvk_mask_trials(hipStream_t st, long n, const unsigned char *trials, double fill, double *scores)
{
    if (n <= 0) return 0;
    const long nb = (n + 255) / 256;
    k_mask_trials<<<(unsigned)(nb > 65536 ? 65536 : nb), 256, 0, st>>>(n, trials, fill, scores);
    return (int)hipGetLastError();
}

// C = beta C + sum_z slab[z]   (split-K epilogue)
__global__ void k_splitk_reduce(int M, int N, int nz, const double *__restrict__ slab, double beta, double *__restrict__ C, long ldc)
{
    const long tot = (long)M * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int z = 0; z < nz; ++z) s += slab[(size_t)z * tot + e];
        const long m = e / N, n = e - m * N;
        C[m * ldc + n] = (beta != 0.0 ? beta * C[m * ldc + n] : 0.0) + s;
    }
}

// Few output tiles, long K: split K over nz = tvk_splitk_count(...) workgroup layers writing slabs
// (nz * M * N doubles of workspace), then reduce.  Deterministic (no atomics).
int tvk_splitk_count(int M, int N, int K, int n_cu)
{
    const long tiles = (long)((N + 127) / 128) * ((M + 127) / 128);
    if (tiles >= 2L * n_cu || K < 2048) return 1;
    long nz = (6L * n_cu + tiles - 1) / tiles; // 3 rounds of the 2 x n_cu workgroup slots: aux (1024 x 400 x 122880) 1.86 -> 1.76 ms against 2 rounds (tools/gemm_probe.hip)
    const long maxz = K / 512 > 1 ? K / 512 : 1;
    if (nz > maxz) nz = maxz;
    return (int)(nz < 1 ? 1 : nz);
}
int tvk_dgemm_splitk(hipStream_t st, bool ta, bool tb, int M, int N, int K, double alpha, const double *A, long lda,
                     const double *B, long ldb, double beta, double *C, long ldc, int nz, double *slabs)
{
    if (M <= 0 || N <= 0) return 0;
    if (nz <= 1) return tvk_dgemm(st, ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
    int kc = ((K + nz - 1) / nz + 15) / 16 * 16;
    nz = (K + kc - 1) / kc;
    dim3 grid((N + 127) / 128, (M + 127) / 128, nz);
    // N = 5 x 80 (the i-vector rank 400) on full row tiles: 80-wide tiles, no strip (option "gemm_nt80", default on)
    const bool nt80 = gmmiv_kopts_cur().gemm_nt80 && !ta && tb && M % 128 == 0 && N % 80 == 0 && N % 128 != 0 && K % 16 == 0 &&
                      (((size_t)A | (size_t)B) % 16 == 0) && lda % 2 == 0 && ldb % 2 == 0;
    if (nt80) launch_dgemm_nt80(st, M, N, K, alpha, A, lda, B, ldb, 0.0, slabs, N, (long)M * N, kc, nz);
    else launch_dgemm(st, ta, tb, grid, M, N, K, alpha, A, lda, 0, B, ldb, 0, 0.0, slabs, N, (long)M * N, kc);
    const long tot = (long)M * N;
    k_splitk_reduce<<<(unsigned)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256), 256, 0, st>>>(M, N, nz, slabs, beta, C, ldc);
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// element-wise / packing kernels
// -------------------------------------------------------------------------------------------
// F[u,c,:] -= mean[c,:] * N[u,c]      (TVAcc::substractM)
__global__ void k_subtract_m(long U, int C, int D, const double *__restrict__ N, double *__restrict__ F,
                             const double *__restrict__ means)
{
    const size_t SV = (size_t)C * D;
    const size_t tot = (size_t)U * SV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t u = e / SV, k = e - u * SV;
        F[e] -= means[k] * N[u * C + k / D];
    }
}

// Fd[u,c,:] = Fs[u,c,:] - mean[c,:] * N[u,c]: restore + substractM in ONE pass (TotalVariability reloads F and centres it at the
// top of every iteration; the copy-then-subtract pair moved the statistics twice).  One thread per 16-byte pair, D even.
__global__ void k_subtract_m_to(long U, int C, int D, const double *__restrict__ N, const double *__restrict__ Fs, double *__restrict__ Fd,
                                const double *__restrict__ means)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const size_t SV2 = (size_t)C * D / 2, tot = (size_t)U * SV2;
    const unsigned D2 = (unsigned)D / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t u = e / SV2;
        const unsigned k2 = (unsigned)(e - u * SV2);
        const double n = N[u * C + k2 / D2];
        const d2 m = *(const d2 *)(means + 2 * (size_t)k2);
        const d2 f = __builtin_nontemporal_load((const d2 *)(Fs + 2 * e));
        d2 o;
        o[0] = f[0] - m[0] * n;
        o[1] = f[1] - m[1] * n;
        *(d2 *)(Fd + 2 * e) = o;
    }
}

// out[i][k] = in[i][k] * scale[k]
__global__ void k_scale_cols(long rows, long cols, const double *__restrict__ in, const double *__restrict__ scale,
                             double *__restrict__ out)
{
    const size_t tot = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x)
        out[e] = in[e] * scale[e % cols];
}

// full[b][i][j] (n x n) <- packed lower [b][i(i+1)/2 + j] mirrored, + diag_add on the diagonal
__global__ void k_unpack_sym(int n, const double *__restrict__ packed, long sp, double *__restrict__ full,
                             double diag_add)
{
    const double *p = packed + (size_t)blockIdx.y * sp;
    double *f = full + (size_t)blockIdx.y * n * n;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e / n, j = e - i * n;
        const int a = i > j ? i : j, b = i > j ? j : i;
        f[e] = p[(size_t)a * (a + 1) / 2 + b] + (i == j ? diag_add : 0.0);
    }
}

// packed[b][i(i+1)/2 + j] <- full[b][i][j] (lower) [+ w[b][i] w[b][j] when w != NULL]
__global__ void k_pack_sym(int n, const double *__restrict__ full, long sf, const double *__restrict__ w,
                           double *__restrict__ packed, long sp)
{
    const double *f = full + (size_t)blockIdx.y * sf;
    double *p = packed + (size_t)blockIdx.y * sp;
    const double *wv = w ? w + (size_t)blockIdx.y * n : nullptr;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e / n, j = e - i * n;
        if (j > i) continue;
        double v = f[e];
        if (wv) v += wv[i] * wv[j];
        p[(size_t)i * (i + 1) / 2 + j] = v;
    }
}

// dst[e] += sum_b src[b*stride + e]   (column sums over a batch)
__global__ void k_batch_sum(long n, int nb, const double *__restrict__ src, long stride, double *__restrict__ dst)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += src[(size_t)b * stride + e];
        dst[e] += s;
    }
}

// two-stage form for long batches: slab y sums its rows [y rows_per, (y + 1) rows_per) into tmp[y][e] (16 x more waves in flight than
// one thread per column walking 1024 rows: the one-stage kernel ran at 1.5 TB/s), then dst[e] += sum_y tmp[y][e] in fixed order
__global__ void k_batch_sum_part(long n, int nb, int rows_per, const double *__restrict__ src, long stride, double *__restrict__ tmp)
{
    const int b0 = blockIdx.y * rows_per, b1 = b0 + rows_per < nb ? b0 + rows_per : nb;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            s0 += src[(size_t)b * stride + e];
            s1 += src[(size_t)(b + 1) * stride + e];
            s2 += src[(size_t)(b + 2) * stride + e];
            s3 += src[(size_t)(b + 3) * stride + e];
        }
        for (; b < b1; ++b) s0 += src[(size_t)b * stride + e];
        tmp[(size_t)blockIdx.y * n + e] = (s0 + s1) + (s2 + s3);
    }
}
__global__ void k_batch_sum_fin(long n, int ny, const double *__restrict__ tmp, double *__restrict__ dst)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int y = 0; y < ny; ++y) s += tmp[(size_t)y * n + e];
        dst[e] += s;
    }
}

// column sums of a NARROW matrix (n of a few hundred columns: sum_u w_u over thousands of utterances) into one or two accumulators.
// k_batch_sum gives such a matrix two workgroups whose threads each walk all nb rows -- 0.63 ms for 2048 x 400, twice per E-step
// (r and meanW both accumulate sum_u w_u), 3 % of a T-matrix iteration.  Here slab y of 64-column group x is a workgroup: its four waves
// take the slab's rows in turn, meet in LDS, and tmp[y][e] gets the slab's sum; k_colsum_narrow_fin adds the slabs in fixed order.
#define TVK_NARROW_SLABS 32
__global__ __launch_bounds__(256) void k_colsum_narrow_part(int n, int nb, int rows_per, const double *__restrict__ src, long stride,
                                                            double *__restrict__ tmp)
{
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e = blockIdx.x * 64 + lane;
    const int b0 = blockIdx.y * rows_per, b1 = b0 + rows_per < nb ? b0 + rows_per : nb;
    double s0 = 0.0, s1 = 0.0;
    if (e < n) {
        int b = b0 + wave;
        for (; b + 4 < b1; b += 8) {
            s0 += src[(size_t)b * stride + e];
            s1 += src[(size_t)(b + 4) * stride + e];
        }
        if (b < b1) s0 += src[(size_t)b * stride + e];
    }
    part[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && e < n) tmp[(size_t)blockIdx.y * n + e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void k_colsum_narrow_fin(int n, int ny, const double *__restrict__ tmp, double *__restrict__ dst, double *__restrict__ dst2)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int y = 0; y < ny; ++y) s += tmp[(size_t)y * n + e];
    dst[e] += s;
    if (dst2) dst2[e] += s;
}

// full symmetric [n x n] += unpack(packed lower [n(n+1)/2])
__global__ void k_add_unpacked(int n, const double *__restrict__ packed, double *__restrict__ full)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e / n, j = e - i * n;
        const int a = i > j ? i : j, b = i > j ? j : i;
        full[e] += packed[(size_t)a * (a + 1) / 2 + b];
    }
}

// y[b][i] = sum_k Mx[b][i][k] x[b][k]   (one workgroup per matrix; n <= a few thousand)
__global__ __launch_bounds__(256) void k_batched_matvec(int n, const double *__restrict__ Mx, const double *__restrict__ x,
                                                        double *__restrict__ y)
{
    const double *m = Mx + (size_t)blockIdx.x * n * n;
    const double *xv = x + (size_t)blockIdx.x * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < n; i += 4) {
        double s = 0.0;
        for (int k = lane; k < n; k += 64) s = __builtin_fma(m[(size_t)i * n + k], xv[k], s);
        s = wave_sum_f64(s);
        if (lane == 0) y[(size_t)blockIdx.x * n + i] = s;
    }
}

// y[j] += sum_k x[k] Mx[k][j]   (Mx: rows x cols; used for mean += T^T meanW)
__global__ void k_vecmat_add(int rows, long cols, const double *__restrict__ x, const double *__restrict__ Mx,
                             double *__restrict__ y)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < (size_t)cols; j += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < rows; ++k) s = __builtin_fma(x[k], Mx[(size_t)k * cols + j], s);
        y[j] += s;
    }
}

// strided 2-D copy: dst[b][i][j] = src[b][i][j], i < rows, j < cols
__global__ void k_copy2d(int rows, int cols, const double *__restrict__ src, long lds_, long ss,
                         double *__restrict__ dst, long ldd, long sd)
{
    const double *s = src + (size_t)blockIdx.y * ss;
    double *d = dst + (size_t)blockIdx.y * sd;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += gridDim.x * blockDim.x) {
        const int i = e / cols, j = e - i * cols;
        d[(size_t)i * ldd + j] = s[(size_t)i * lds_ + j];
    }
}

// -------------------------------------------------------------------------------------------
// Diagonal block of the blocked Cholesky: factor the w x w block (w <= 32) in LDS, write the
// lower factor back (upper part zeroed) and its inverse to invd[b][kb][32][32].
// status[b] = 1 when a non-positive pivot is met.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potf2_inv(int n, int kk, int w, double *__restrict__ Afull,
                                                   double *__restrict__ invd, long sinv, int kb, int *__restrict__ status)
{
    __shared__ double a[32][33];
    __shared__ double x[32][33];
    __shared__ int bad;
    double *A = Afull + (size_t)blockIdx.x * n * n + (size_t)kk * n + kk;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        a[i][j] = (i < w && j < w) ? A[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
        x[i][j] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < w; ++j) {
        if (tid == 0) {
            double d = a[j][j];
            if (!(d > 0.0)) { bad = 1; d = 1.0; }
            a[j][j] = sqrt(d);
        }
        __syncthreads();
        const double dj = a[j][j];
        if (tid > j && tid < w) a[tid][j] /= dj;
        __syncthreads();
        // trailing update of the lower triangle: a[i][k] -= a[i][j] a[k][j], j < k <= i
        for (int e = tid; e < 32 * 32; e += 256) {
            const int i = e >> 5, k = e & 31;
            if (k > j && k <= i && i < w) a[i][k] -= a[i][j] * a[k][j];
        }
        __syncthreads();
    }
    // inverse of the lower-triangular factor, one column per thread (forward substitution)
    if (tid < w) {
        const int c = tid;
        x[c][c] = 1.0 / a[c][c];
        for (int i = c + 1; i < w; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s = __builtin_fma(a[i][k], x[k][c], s);
            x[i][c] = -s / a[i][i];
        }
    }
    __syncthreads();
    double *iv = invd + (size_t)blockIdx.x * sinv + (size_t)kb * 1024;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, j = e & 31;
        if (i < w && j < w) A[(size_t)i * n + j] = (j <= i) ? a[i][j] : 0.0;
        iv[e] = (i < w && j < w && j <= i) ? x[i][j] : 0.0;
    }
    if (tid == 0 && bad) status[blockIdx.x] = 1;
}

#define TVCHK(e)                     \
    do {                             \
        int _r = (e);                \
        if (_r) return _r;           \
    } while (0)

#define g_chol_gemm_path (gmmiv_kopts_cur().chol_gemm) // A/B knob of the calling context: 1 = the GEMM-built right-looking factorisation for every n
int tvk_chol_accepts_packed(int n) { return n % 2 == 0 && !g_chol_gemm_path; }

// In-place batched Cholesky (lower) of nb SPD matrices [n x n]; invd receives the inverses of the
// 32 x 32 diagonal blocks; panel: scratch nb*n*32 doubles. Upper triangles end up unspecified
// except inside diagonal blocks (zeroed).
int tvk_chol_batched(hipStream_t st, int n, int nb, double *Afull, double *invd, double *panel, int *status)
{
    // Two-level right-looking factorisation: 32-wide steps (diagonal block factored and inverted in LDS, panel
    // by GEMM) inside 128-wide outer blocks.  A step only updates the columns of its own outer block; the rest of
    // the trailing matrix is updated once per outer block with K = 128 -- a quarter of the read-modify-write
    // traffic of updating the whole trailing matrix at every step (the batch is memory-bound there).
    if (n % 2 == 0 && !g_chol_gemm_path) return tvk_chol_left_batched(st, n, nb, Afull, invd, status);
    constexpr int WB = 128;
    const int nblk = (n + 31) / 32;
    const long sinv = (long)nblk * 1024, sa = (long)n * n, spn = (long)n * 32;
    for (int j0 = 0; j0 < n; j0 += WB) {
        const int jend = j0 + WB < n ? j0 + WB : n;
        for (int kk = j0; kk < jend; kk += 32) {
            const int kb = kk / 32, w = (n - kk) < 32 ? (n - kk) : 32, m = n - kk - w;
            k_potf2_inv<<<nb, 256, 0, st>>>(n, kk, w, Afull, invd, sinv, kb, status);
            if (m > 0) {
                // L21 = A21 * inv(L11)^T
                TVCHK(tvk_dgemm(st, false, true, m, w, w, 1.0, Afull + (size_t)(kk + w) * n + kk, n, sa, invd + (size_t)kb * 1024,
                                32, sinv, 0.0, panel, 32, spn, nb));
                dim3 g((m * w + 255) / 256, nb);
                k_copy2d<<<g, 256, 0, st>>>(m, w, panel, 32, spn, Afull + (size_t)(kk + w) * n + kk, n, sa);
                // columns kk+w .. jend of the rows below: A22[:, :cin] -= L21 * L21[0:cin]^T
                const int cin = jend - (kk + w);
                if (cin > 0)
                    TVCHK(tvk_dgemm(st, false, true, m, cin, w, -1.0, panel, 32, spn, panel, 32, spn, 1.0,
                                    Afull + (size_t)(kk + w) * n + (kk + w), n, sa, nb));
            }
        }
        const int m2 = n - jend;
        if (m2 > 0) { // trailing matrix beyond the outer block: -= L[jend:, j0:jend] * L[jend:, j0:jend]^T
            const double *Lp = Afull + (size_t)jend * n + j0;
            TVCHK(tvk_dgemm(st, false, true, m2, m2, jend - j0, -1.0, Lp, n, sa, Lp, n, sa, 1.0, Afull + (size_t)jend * n + jend, n, sa, nb));
        }
    }
    return (int)hipGetLastError();
}

// inv[b] = A[b]^-1 for nb SPD matrices; A is destroyed (holds the Cholesky factor afterwards).
// X: scratch nb*n*n, invd: nb*nblk*1024, panel: nb*n*32 doubles.
int tvk_spd_inverse_batched(hipStream_t st, int n, int nb, double *Afull, double *inv, double *X, double *invd,
                            double *panel, int *status)
{
    if (n % 2 == 0 && !g_chol_gemm_path) return tvk_spd_inverse_left_batched(st, n, nb, Afull, inv, X, invd, status);
    TVCHK(tvk_chol_batched(st, n, nb, Afull, invd, panel, status));
    const int nblk = (n + 31) / 32;
    const long sinv = (long)nblk * 1024, sa = (long)n * n, spn = (long)n * 32;
    if (hipMemsetAsync(X, 0, (size_t)nb * sa * sizeof(double), st) != hipSuccess) return (int)hipGetLastError();
    for (int ib = 0; ib < nblk; ++ib) {
        const int r0 = ib * 32, w = (n - r0) < 32 ? (n - r0) : 32;
        dim3 g((w * w + 255) / 256, nb);
        k_copy2d<<<g, 256, 0, st>>>(w, w, invd + (size_t)ib * 1024, 32, sinv, X + (size_t)r0 * n + r0, n, sa);
        if (ib > 0) {
            // tmp = L[ib, 0:r0] * X[0:r0, 0:r0]          (panel reused as [32 x n] scratch, ld = n)
            TVCHK(tvk_dgemm(st, false, false, w, r0, r0, 1.0, Afull + (size_t)r0 * n, n, sa, X, n, sa, 0.0, panel, n, spn, nb));
            // X[ib, 0:r0] = -inv(L_ii) * tmp
            TVCHK(tvk_dgemm(st, false, false, w, r0, w, -1.0, invd + (size_t)ib * 1024, 32, sinv, panel, n, spn, 0.0,
                            X + (size_t)r0 * n, n, sa, nb));
        }
    }
    // A^-1 = X^T X
    TVCHK(tvk_dgemm(st, true, false, n, n, n, 1.0, X, n, sa, X, n, sa, 0.0, inv, n, sa, nb));
    return (int)hipGetLastError();
}

// w = A^-1 b through the Cholesky factor (lower, from tvk_chol_batched) and the inverses of its
// 32 x 32 diagonal blocks: blocked forward then backward substitution, one workgroup per matrix.
// 256 threads = 32 rows (or columns) x 8 partial-sum lanes.
__global__ __launch_bounds__(256) void k_chol_solve(int n, const double *__restrict__ Lf, const double *__restrict__ invd,
                                                    long sinv, const double *__restrict__ bvec, double *__restrict__ wvec)
{
    extern __shared__ double sm[];      // y[npad] | r[32]
    const int nblk = (n + 31) / 32, npad = nblk * 32;
    double *y = sm, *rr = sm + npad;
    const double *L = Lf + (size_t)blockIdx.x * n * n;
    const double *iv = invd + (size_t)blockIdx.x * sinv;
    const double *bb = bvec + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, row = tid >> 3, l8 = tid & 7;
    for (int i = tid; i < npad; i += 256) y[i] = 0.0;
    __syncthreads();
    // forward: L y = b
    for (int ib = 0; ib < nblk; ++ib) {
        const int r0 = ib * 32, gr = r0 + row;
        double s = 0.0;
        if (gr < n)
            for (int k = l8; k < r0; k += 8) s = __builtin_fma(L[(size_t)gr * n + k], y[k], s);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (l8 == 0) rr[row] = gr < n ? bb[gr] - s : 0.0;
        __syncthreads();
        double t = 0.0;
        for (int k = l8; k < 32; k += 8) t = __builtin_fma(iv[(size_t)ib * 1024 + row * 32 + k], rr[k], t); // inv(L_ii) is lower
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        __syncthreads();
        if (l8 == 0) y[r0 + row] = t;
        __syncthreads();
    }
    // backward: L^T w = y   (w overwrites y block by block, from the last block up)
    for (int ib = nblk - 1; ib >= 0; --ib) {
        const int c0 = ib * 32, gc = c0 + row; // `row` indexes a column of the block here
        double s = 0.0;
        if (gc < n)
            for (int k = c0 + 32 + l8; k < n; k += 8) s = __builtin_fma(L[(size_t)k * n + gc], y[k], s);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (l8 == 0) rr[row] = gc < n ? y[gc] - s : 0.0;
        __syncthreads();
        double t = 0.0;
        for (int k = l8; k < 32; k += 8) t = __builtin_fma(iv[(size_t)ib * 1024 + k * 32 + row], rr[k], t); // inv(L_ii)^T
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        __syncthreads();
        if (l8 == 0) y[c0 + row] = t;
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) wvec[(size_t)blockIdx.x * n + i] = y[i];
}

int tvk_chol_solve_batched(hipStream_t st, int n, int nb, const double *Lf, const double *invd, const double *b, double *w)
{
    if (nb <= 0) return 0;
    const int nblk = (n + 31) / 32;
    k_chol_solve<<<nb, 256, (size_t)(nblk * 32 + 32) * sizeof(double), st>>>(n, Lf, invd, (long)nblk * 1024, b, w);
    return (int)hipGetLastError();
}

// ---- approximate extractors (normStatistics, substractMplusTW, normTMatrix, getWeightedCov, approximateTcTc) ----
// F[u,c,d] = (F - mean[c,d] N[u,c]) * sqrt(invvar[c,d])
__global__ void k_norm_stats(long U, int C, int D, const double *__restrict__ N, double *__restrict__ F,
                             const double *__restrict__ means, const double *__restrict__ invvar)
{
    const long SV = (long)C * D, n = U * SV;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long u = e / SV, k = e - u * SV;
        F[e] = (F[e] - means[k] * N[u * C + k / D]) * sqrt(invvar[k]);
    }
}
// F[u,c,d] -= (mean[c,d] + TW[u, cD+d]) * N[u,c]
__global__ void k_sub_mtw(long U, int C, int D, const double *__restrict__ N, double *__restrict__ F,
                          const double *__restrict__ means, const double *__restrict__ TW)
{
    const long SV = (long)C * D, n = U * SV;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long u = e / SV, k = e - u * SV;
        F[e] -= (means[k] + TW[e]) * N[u * C + k / D];
    }
}
// out[r][k] = in[r][k] * (mode 0: sqrt(v[k]); mode 1: v[k / D])
__global__ void k_scale_cols_fn(long rows, long cols, int D, int mode, const double *__restrict__ in,
                                const double *__restrict__ v, double *__restrict__ out)
{
    const long n = rows * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long k = e % cols;
        out[e] = in[e] * (mode == 0 ? sqrt(v[k]) : v[k / D]);
    }
}
// Dm[c,i] += sum_k A[(cD+k), i]^2   (A is [SV x R])
__global__ void k_block_colnorm(int C, int D, int R, const double *__restrict__ A, double *__restrict__ Dm)
{
    const long n = (long)C * R;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long c = e / R, i = e - c * R;
        double s = 0.0;
        for (int k = 0; k < D; ++k) { const double a = A[((size_t)c * D + k) * R + i]; s = __builtin_fma(a, a, s); }
        Dm[e] += s;
    }
}
// full[b] = I + (sum_c N[b,c]) * Wm   (one workgroup per matrix)
__global__ __launch_bounds__(256) void k_build_l_ubm(int R, int C, const double *__restrict__ N, const double *__restrict__ Wm,
                                                     double *__restrict__ full)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) s += N[(size_t)blockIdx.x * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const double ns = red[0];
    double *L = full + (size_t)blockIdx.x * R * R;
    for (int e = threadIdx.x; e < R * R; e += 256) L[e] = ns * Wm[e] + ((e / R) == (e % R) ? 1.0 : 0.0);
}
// B[e] *= 1 / (1 + X[e])
__global__ void k_mul_recip1p(long n, double *__restrict__ B, const double *__restrict__ X)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) B[e] *= 1.0 / (1.0 + X[e]);
}
__global__ void k_add_identity(int n, double *__restrict__ A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * n + i] += 1.0;
}

// ---- development-set statistics (PldaDev): X [dim x n], one vector per column, sessions grouped by speaker ----
// ssum[k, c] = sum of row k over the sessions of speaker c (off[c] .. off[c+1])
__global__ void k_dev_spk_sums(int dim, long n, const double *__restrict__ X, long nspk, const long *__restrict__ off,
                               double *__restrict__ ssum)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)dim * nspk) return;
    const long k = e / nspk, c = e - k * nspk;
    double s = 0.0;
    for (long j = off[c]; j < off[c + 1]; ++j) s += X[k * n + j];
    ssum[e] = s;
}
// mean[k] = sum_c ssum[k,c] / n ; smean[k,c] = ssum[k,c] / count_c.  One workgroup per dimension k, its threads stride over the
// speakers (coalesced; the first version gave each of the `dim` threads a whole row to walk: 0.63 ms for 400 x 20 k speakers),
// partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void k_dev_means(int dim, long n, long nspk, const long *__restrict__ off, const double *__restrict__ ssum,
                                                   double *__restrict__ mean, double *__restrict__ smean)
{
    __shared__ double part[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (long c = tid; c < nspk; c += 256) {
        const double v = ssum[(size_t)k * nspk + c];
        s += v;
        smean[(size_t)k * nspk + c] = v / (double)(off[c + 1] - off[c]);
    }
    part[tid] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) part[tid] += part[tid + h];
        __syncthreads();
    }
    if (tid == 0) mean[k] = part[0] / (double)n;
}
// mode 0: out = X - mean ; 1: out = X - smean[class] ; 2: out = (X - smean[class]) / sqrt(count[class])
__global__ void k_dev_center(int dim, long n, int mode, const double *__restrict__ X, const double *__restrict__ mean,
                             const double *__restrict__ smean, long nspk, const long *__restrict__ off, const int *__restrict__ cls,
                             double *__restrict__ out)
{
    const long tot = (long)dim * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long k = e / n, s = e - k * n;
        if (mode == 0) out[e] = X[e] - mean[k];
        else {
            const int c = cls[s];
            const double v = X[e] - smean[(size_t)k * nspk + c];
            out[e] = mode == 1 ? v : v / sqrt((double)(off[c + 1] - off[c]));
        }
    }
}
// out[k,c] = (smean[k,c] - mean[k]) * (weighted ? sqrt(count_c) : 1)
__global__ void k_dev_between(int dim, long nspk, int weighted, const double *__restrict__ mean, const double *__restrict__ smean,
                              const long *__restrict__ off, double *__restrict__ out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)dim * nspk) return;
    const long k = e / nspk, c = e - k * nspk;
    out[e] = (smean[e] - mean[k]) * (weighted ? sqrt((double)(off[c + 1] - off[c])) : 1.0);
}
// out[r, s] = H[r, cls[s]]   (H [rows x nspk])
__global__ void k_dev_expand(int rows, long n, long nspk, const double *__restrict__ H, const int *__restrict__ cls, double *__restrict__ out)
{
    const long tot = (long)rows * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long r = e / n, s = e - r * n;
        out[e] = H[r * nspk + cls[s]];
    }
}
int tvk_dev_means(hipStream_t st, int dim, long n, const double *X, long nspk, const long *off, double *ssum, double *mean, double *smean)
{
    k_dev_spk_sums<<<(unsigned)(((long)dim * nspk + 255) / 256), 256, 0, st>>>(dim, n, X, nspk, off, ssum);
    k_dev_means<<<dim, 256, 0, st>>>(dim, n, nspk, off, ssum, mean, smean);
    return (int)hipGetLastError();
}
static unsigned ew_blocks(long n);
int tvk_dev_center(hipStream_t st, int dim, long n, int mode, const double *X, const double *mean, const double *smean, long nspk,
                   const long *off, const int *cls, double *out)
{
    if (n <= 0) return 0;
    k_dev_center<<<ew_blocks((long)dim * n), 256, 0, st>>>(dim, n, mode, X, mean, smean, nspk, off, cls, out);
    return (int)hipGetLastError();
}
int tvk_dev_expand(hipStream_t st, int rows, long n, long nspk, const double *H, const int *cls, double *out)
{
    if (n <= 0) return 0;
    k_dev_expand<<<ew_blocks((long)rows * n), 256, 0, st>>>(rows, n, nspk, H, cls, out);
    return (int)hipGetLastError();
}
int tvk_dev_between(hipStream_t st, int dim, long nspk, int weighted, const double *mean, const double *smean, const long *off, double *out)
{
    k_dev_between<<<(unsigned)(((long)dim * nspk + 255) / 256), 256, 0, st>>>(dim, nspk, weighted, mean, smean, off, out);
    return (int)hipGetLastError();
}

static unsigned ew_blocks(long n) { long b = (n + 255) / 256; return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }
int tvk_norm_stats(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means, const double *invvar)
{
    if (U <= 0) return 0;
    k_norm_stats<<<ew_blocks(U * C * D), 256, 0, st>>>(U, C, D, N, F, means, invvar);
    return (int)hipGetLastError();
}
int tvk_sub_mtw(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means, const double *TW)
{
    if (U <= 0) return 0;
    k_sub_mtw<<<ew_blocks(U * C * D), 256, 0, st>>>(U, C, D, N, F, means, TW);
    return (int)hipGetLastError();
}
int tvk_scale_cols_fn(hipStream_t st, long rows, long cols, int D, int mode, const double *in, const double *v, double *out)
{
    k_scale_cols_fn<<<ew_blocks(rows * cols), 256, 0, st>>>(rows, cols, D, mode, in, v, out);
    return (int)hipGetLastError();
}
int tvk_block_colnorm(hipStream_t st, int C, int D, int R, const double *A, double *Dm)
{
    k_block_colnorm<<<ew_blocks((long)C * R), 256, 0, st>>>(C, D, R, A, Dm);
    return (int)hipGetLastError();
}
int tvk_build_l_ubm(hipStream_t st, int R, int C, int nb, const double *N, const double *Wm, double *full)
{
    if (nb <= 0) return 0;
    k_build_l_ubm<<<nb, 256, 0, st>>>(R, C, N, Wm, full);
    return (int)hipGetLastError();
}
int tvk_mul_recip1p(hipStream_t st, long n, double *B, const double *X)
{
    if (n <= 0) return 0;
    k_mul_recip1p<<<ew_blocks(n), 256, 0, st>>>(n, B, X);
    return (int)hipGetLastError();
}
int tvk_add_identity(hipStream_t st, int n, double *A)
{
    k_add_identity<<<(n + 255) / 256, 256, 0, st>>>(n, A);
    return (int)hipGetLastError();
}

// ---- JFA (AccumulateJFAStat.cpp): statistics minus the model terms, diagonal factor z / D ---------------------------
// Wg[r][:] = W[owner ? owner[r0 + r] : r0 + r][:]
__global__ void k_gather_rows(int nb, int R, long r0, const long *__restrict__ owner, const double *__restrict__ W, double *__restrict__ Wg)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)nb * R; e += (long)gridDim.x * blockDim.x) {
        const long r = e / R, j = e - r * R;
        const long o = owner ? owner[r0 + r] : r0 + r;
        Wg[e] = W[o * R + j];
    }
}
// F[r,c,:] -= N[r,c] (means[c,:] + TW[r - r0][c,:] + Dm[c,:] Z[owner(r)][c,:])   (every term optional)
__global__ void k_jfa_sub(long nb, int C, int D, long r0, const long *__restrict__ owner, const double *__restrict__ N,
                          double *__restrict__ F, const double *__restrict__ means, const double *__restrict__ TW,
                          const double *__restrict__ Dm, const double *__restrict__ Z)
{
    const size_t SV = (size_t)C * D, tot = (size_t)nb * SV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t rl = e / SV, k = e - rl * SV, r = r0 + rl;
        double v = means ? means[k] : 0.0;
        if (TW) v += TW[e];
        if (Dm) v = __builtin_fma(Dm[k], Z[(size_t)(owner ? owner[r] : (long)r) * SV + k], v);
        F[r * SV + k] -= N[r * C + k / D] * v;
    }
}
// F_X[s,c,:] -= sum over the sessions h of speaker s inside [h0, h1) of N_h[h,c] G[h - h0][c,:]     (G = X U)
__global__ void k_jfa_sub_sessions(long s0, long ns, long h0, long h1, int C, int D, const long *__restrict__ sess_begin,
                                   const double *__restrict__ Nh, const double *__restrict__ G, double *__restrict__ FX)
{
    const size_t SV = (size_t)C * D, tot = (size_t)ns * SV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t sl = e / SV, k = e - sl * SV, s = s0 + sl;
        long a = sess_begin[s], b = sess_begin[s + 1];
        a = a > h0 ? a : h0;
        b = b < h1 ? b : h1;
        double acc = 0.0;
        for (long h = a; h < b; ++h) acc = __builtin_fma(Nh[h * C + k / D], G[(size_t)(h - h0) * SV + k], acc);
        if (b > a) FX[s * SV + k] -= acc;
    }
}
// tau < 0: z = F iv D / (1 + N iv D^2) (estimateZ, :3550-3573);  tau >= 0: z = tau / (tau + N) D iv F (estimateZMAP, :3576-3594)
__global__ void k_jfa_z(long nspk, int C, int D, const double *__restrict__ N, const double *__restrict__ F,
                        const double *__restrict__ iv, const double *__restrict__ Dm, double tau, double *__restrict__ Z)
{
    const size_t SV = (size_t)C * D, tot = (size_t)nspk * SV;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t s = e / SV, k = e - s * SV;
        const double n = N[s * C + k / D], d = Dm[k], v = iv[k];
        Z[e] = tau < 0.0 ? F[e] * v * d / (1.0 + n * v * d * d) : (tau / (tau + n)) * d * v * F[e];
    }
}
// estimateZandD (:3480-3516): z as above, D[k] = sum_s z F / sum_s (1 / L + z^2) N; one thread per supervector entry,
// speakers in order (deterministic)
__global__ void k_jfa_z_and_d(long nspk, int C, int D, const double *__restrict__ N, const double *__restrict__ F,
                              const double *__restrict__ iv, double *__restrict__ Dm, double *__restrict__ Z)
{
    const size_t SV = (size_t)C * D;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= SV) return;
    const double d = Dm[k], v = iv[k];
    double a1 = 0.0, a2 = 0.0;
    for (long s = 0; s < nspk; ++s) {
        const double n = N[s * C + k / D], f = F[s * SV + k];
        const double L = 1.0 + n * v * d * d;
        const double z = f * v * d / L;
        Z[s * SV + k] = z;
        a1 += (1.0 / L + z * z) * n;
        a2 += z * f;
    }
    Dm[k] = a2 / a1;
}
int tvk_gather_rows(hipStream_t st, int nb, int R, long r0, const long *owner, const double *W, double *Wg)
{
    if (nb <= 0) return 0;
    k_gather_rows<<<ew_blocks((long)nb * R), 256, 0, st>>>(nb, R, r0, owner, W, Wg);
    return (int)hipGetLastError();
}
int tvk_jfa_sub(hipStream_t st, long nb, int C, int D, long r0, const long *owner, const double *N, double *F, const double *means,
                const double *TW, const double *Dm, const double *Z)
{
    if (nb <= 0) return 0;
    k_jfa_sub<<<ew_blocks(nb * C * D), 256, 0, st>>>(nb, C, D, r0, owner, N, F, means, TW, Dm, Z);
    return (int)hipGetLastError();
}
int tvk_jfa_sub_sessions(hipStream_t st, long s0, long ns, long h0, long h1, int C, int D, const long *sess_begin, const double *Nh,
                         const double *G, double *FX)
{
    if (ns <= 0) return 0;
    k_jfa_sub_sessions<<<ew_blocks(ns * C * D), 256, 0, st>>>(s0, ns, h0, h1, C, D, sess_begin, Nh, G, FX);
    return (int)hipGetLastError();
}
int tvk_jfa_z(hipStream_t st, long nspk, int C, int D, const double *N, const double *F, const double *iv, const double *Dm, double tau, double *Z)
{
    if (nspk <= 0) return 0;
    k_jfa_z<<<ew_blocks(nspk * C * D), 256, 0, st>>>(nspk, C, D, N, F, iv, Dm, tau, Z);
    return (int)hipGetLastError();
}
int tvk_jfa_z_and_d(hipStream_t st, long nspk, int C, int D, const double *N, const double *F, const double *iv, double *Dm, double *Z)
{
    const long SV = (long)C * D;
    k_jfa_z_and_d<<<(unsigned)((SV + 255) / 256), 256, 0, st>>>(nspk, C, D, N, F, iv, Dm, Z);
    return (int)hipGetLastError();
}

int tvk_subtract_m(hipStream_t st, long U, int C, int D, const double *N, double *F, const double *means)
{
    if (U <= 0) return 0;
    k_subtract_m<<<2048, 256, 0, st>>>(U, C, D, N, F, means);
    return (int)hipGetLastError();
}
int tvk_subtract_m_to(hipStream_t st, long U, int C, int D, const double *N, const double *Fs, double *Fd, const double *means)
{
    if (U <= 0) return 0;
    if ((D & 1) || (((uintptr_t)Fs | (uintptr_t)Fd | (uintptr_t)means) & 15)) return -1; // the caller copies + subtracts in place
    k_subtract_m_to<<<4096, 256, 0, st>>>(U, C, D, N, Fs, Fd, means);
    return (int)hipGetLastError();
}
int tvk_scale_cols(hipStream_t st, long rows, long cols, const double *in, const double *scale, double *out)
{
    k_scale_cols<<<2048, 256, 0, st>>>(rows, cols, in, scale, out);
    return (int)hipGetLastError();
}
int tvk_unpack_sym(hipStream_t st, int n, int nb, const double *packed, long sp, double *full, double diag_add)
{
    if (nb <= 0) return 0;
    dim3 g((n * n + 255) / 256 > 512 ? 512 : (n * n + 255) / 256, nb);
    k_unpack_sym<<<g, 256, 0, st>>>(n, packed, sp, full, diag_add);
    return (int)hipGetLastError();
}
int tvk_pack_sym(hipStream_t st, int n, int nb, const double *full, long sf, const double *w, double *packed, long sp)
{
    if (nb <= 0) return 0;
    dim3 g((n * n + 255) / 256 > 512 ? 512 : (n * n + 255) / 256, nb);
    k_pack_sym<<<g, 256, 0, st>>>(n, full, sf, w, packed, sp);
    return (int)hipGetLastError();
}
// dst[d][j] = sum over p in [off[d], off[d + 1]) of src[rows[p]][j]   (statistics rows of ndx lines = sums of their files' rows;
// fixed order: deterministic)
__global__ void k_merge_rows(long ndst, long width, const long *__restrict__ off, const long *__restrict__ rows,
                             const double *__restrict__ src, double *__restrict__ dst)
{
    const long d = blockIdx.y;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < width; j += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (long p = off[d]; p < off[d + 1]; ++p) s += src[rows[p] * width + j];
        dst[d * width + j] = s;
    }
}
int tvk_merge_rows(hipStream_t st, long ndst, long width, const long *off, const long *rows, const double *src, double *dst)
{
    if (ndst <= 0 || width <= 0) return 0;
    const long bx = (width + 255) / 256;
    for (long d0 = 0; d0 < ndst; d0 += 65535) {
        const long nd = ndst - d0 < 65535 ? ndst - d0 : 65535;
        k_merge_rows<<<dim3((unsigned)(bx > 64 ? 64 : bx), (unsigned)nd), 256, 0, st>>>(nd, width, off + d0, rows, src, dst + d0 * width);
    }
    return (int)hipGetLastError();
}
int tvk_batch_sum(hipStream_t st, long n, int nb, const double *src, long stride, double *dst, double *tmp)
{
    if (nb <= 0 || n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (tmp && nb >= 128 && n >= 4096) { // tmp: TVK_BATCH_SUM_SLABS * n doubles
        const int rows_per = (nb + TVK_BATCH_SUM_SLABS - 1) / TVK_BATCH_SUM_SLABS, ny = (nb + rows_per - 1) / rows_per;
        k_batch_sum_part<<<dim3(blocks, ny), 256, 0, st>>>(n, nb, rows_per, src, stride, tmp);
        k_batch_sum_fin<<<blocks, 256, 0, st>>>(n, ny, tmp, dst);
        return (int)hipGetLastError();
    }
    k_batch_sum<<<blocks, 256, 0, st>>>(n, nb, src, stride, dst);
    return (int)hipGetLastError();
}
// dst[e] += sum_b src[b * stride + e] (and dst2, when given) for a narrow matrix; tmp: TVK_NARROW_SLABS * n doubles
int tvk_colsum_narrow(hipStream_t st, int n, int nb, const double *src, long stride, double *dst, double *dst2, double *tmp)
{
    if (nb <= 0 || n <= 0) return 0;
    if (!tmp || nb < 64) {
        k_batch_sum<<<(n + 255) / 256, 256, 0, st>>>(n, nb, src, stride, dst);
        if (dst2) k_batch_sum<<<(n + 255) / 256, 256, 0, st>>>(n, nb, src, stride, dst2);
        return (int)hipGetLastError();
    }
    const int rows_per = (nb + TVK_NARROW_SLABS - 1) / TVK_NARROW_SLABS, ny = (nb + rows_per - 1) / rows_per;
    k_colsum_narrow_part<<<dim3((n + 63) / 64, ny), 256, 0, st>>>(n, nb, rows_per, src, stride, tmp);
    k_colsum_narrow_fin<<<(n + 255) / 256, 256, 0, st>>>(n, ny, tmp, dst, dst2);
    return (int)hipGetLastError();
}
// estimateTETt (AccumulateTVStat.cpp:777-805): TETt_c = T_c diag(iv_c) T_c^T for every Gaussian, written as PACKED lower rows.
// One workgroup per Gaussian (and pass): the rows [jb, jb + JH) of T_c sit in LDS as the B operand (row stride 4 KS + 2 doubles:
// 8-byte reads of 16 rows x 2 k are conflict-free); a wave keeps the 16 rows of its i tile, scaled by iv, as the A operand in KS
// registers and walks the j tiles of the pass (j <= i) with KS MFMAs each, B operands of the next tile requested under the MFMAs of the
// current one; a 16 x 16 result goes straight to its packed place (16 lanes = 128 contiguous bytes per row).  Only the lower
// triangle is computed and nothing is written twice: the batched-GEMM form wrote 2048 full 400 x 400 matrices (2.6 GB), read them
// back and packed them -- 2.6 ms per iteration where the arithmetic is 0.26 ms and the packed result 1.3 GB.
template <int KS>
__global__ __launch_bounds__(256) void k_tett_packed(int R, int D, long SV, int JH, const double *__restrict__ T, const double *__restrict__ iv,
                                                     double *__restrict__ out, long P)
{
    constexpr int RS = 4 * KS + 2;
    extern __shared__ __attribute__((aligned(16))) double lds_b[]; // [JH][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const long c = blockIdx.x;
    const double *Tc = T + c * D;
    double *oc = out + c * P;
    // this lane's inverse variances: k = 4 s + q
    double ivq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { const int k = 4 * s + q; ivq[s] = k < D ? iv[c * D + k] : 0.0; }
    const int nt = (R + 15) >> 4;
    for (int jb = 0; jb < R; jb += JH) {
        const int jrows = (R - jb) < JH ? (R - jb) : JH;
        __syncthreads(); // the previous pass no longer reads the buffer
        for (int e = tid; e < jrows * (4 * KS); e += 256) {
            const int r = e / (4 * KS), k = e - r * (4 * KS);
            lds_b[r * RS + k] = k < D ? Tc[(long)(jb + r) * SV + k] : 0.0;
        }
        __syncthreads();
        const int jt0 = jb >> 4, jt1 = (jb + jrows + 15) >> 4; // j tiles of this pass
        for (int it = jt0 + wave; it < nt; it += 4) {
            // A operand: rows it * 16 + i16 (clamped), scaled
            long ri = (long)it * 16 + i16;
            ri = ri < R ? ri : R - 1;
            double a[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) { const int k = 4 * s + q; a[s] = k < D ? Tc[ri * SV + k] * ivq[s] : 0.0; }
            const int jend = it + 1 < jt1 ? it + 1 : jt1; // j tiles jt0 .. jend - 1 (j <= i)
            double b0[KS], b1[KS];
            auto bload = [&](double (&b)[KS], int jt) __attribute__((always_inline)) {
                int rj = jt * 16 + i16 - jb;
                rj = rj < jrows ? rj : jrows - 1;
                const double *pb = lds_b + rj * RS + q;
#pragma unroll
                for (int s = 0; s < KS; ++s) b[s] = pb[4 * s];
            };
            auto tile = [&](const double (&b)[KS], int jt) __attribute__((always_inline)) {
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = MFMA_F64(a[s], b[s], acc);
                // D layout: lane holds rows q + 4 r, column i16 of the 16 x 16 tile
                const long col = (long)jt * 16 + i16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = (long)it * 16 + q + 4 * r;
                    if (row < R && col <= row) oc[row * (row + 1) / 2 + col] = acc[r];
                }
            };
            if (jt0 < jend) bload(b0, jt0);
            for (int jt = jt0; jt < jend; jt += 2) {
                if (jt + 1 < jend) bload(b1, jt + 1);
                tile(b0, jt);
                if (jt + 1 < jend) {
                    if (jt + 2 < jend) bload(b0, jt + 2);
                    tile(b1, jt + 1);
                }
            }
        }
    }
}
// 0 = done; -1 = shape outside this kernel (D > 64): the caller runs the batched-GEMM form
int tvk_tett_packed(hipStream_t st, int C, int D, int R, const double *T, const double *iv, double *out)
{
    if (C <= 0 || R <= 0) return 0;
    if (D > 64 || D <= 0) return -1;
    const long SV = (long)C * D, P = (long)R * (R + 1) / 2;
    const int KS = (D + 3) / 4, RS = 4 * KS + 2;
    // rows per pass: two workgroups per CU (<= 72 KB each) unless the whole matrix fits a little above that
    int JH = (72 * 1024) / (RS * 8) / 16 * 16;
    if (JH < 16) JH = 16;
    if (JH > R) JH = (R + 15) / 16 * 16;
    const size_t lds = (size_t)JH * RS * 8;
#define TETT_CASE(K)                                                                                                                  \
    case K: {                                                                                                                         \
        if (gmmiv_lds_attr<k_tett_packed<K>>(lds) != hipSuccess) return -2; /* per (device, kernel): lds_attr.h */                                                     \
        k_tett_packed<K><<<C, 256, lds, st>>>(R, D, SV, JH, T, iv, out, P);                                                           \
    } break;
    switch (KS) {
        TETT_CASE(1) TETT_CASE(2) TETT_CASE(3) TETT_CASE(4) TETT_CASE(5) TETT_CASE(6) TETT_CASE(7) TETT_CASE(8) TETT_CASE(9) TETT_CASE(10)
        TETT_CASE(11) TETT_CASE(12) TETT_CASE(13) TETT_CASE(14) TETT_CASE(15) TETT_CASE(16)
    default: return -1;
    }
#undef TETT_CASE
    return (int)hipGetLastError();
}
int tvk_add_unpacked(hipStream_t st, int n, const double *packed, double *full)
{
    k_add_unpacked<<<(n * n + 255) / 256, 256, 0, st>>>(n, packed, full);
    return (int)hipGetLastError();
}
// minDivergence on the device: Rn = Rm / n - (r / n)(r / n)^T into BOTH Rm (the caller's normalised R) and work (factored in place);
// r /= n by a second launch (every element above reads the old r)
__global__ void k_md_normalize(int R, double inv_n, double *__restrict__ Rm, const double *__restrict__ r, double *__restrict__ work)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * R) return;
    const int i = e / R, j = e - i * R;
    const double v = Rm[e] * inv_n - (r[i] * inv_n) * (r[j] * inv_n);
    Rm[e] = v;
    work[e] = v;
}
__global__ void k_scale_vec(int n, double f, double *__restrict__ v)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) v[e] *= f;
}
// U = L^T of a lower factor stored row-major (whatever sits above L's diagonal is ignored), zero below the diagonal
__global__ void k_lower_to_upper(int n, const double *__restrict__ L, double *__restrict__ U)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    U[e] = j >= i ? L[(size_t)j * n + i] : 0.0;
}
int tvk_md_normalize(hipStream_t st, int R, double n_sessions, double *Rm, double *r, double *work)
{
    k_md_normalize<<<(R * R + 255) / 256, 256, 0, st>>>(R, 1.0 / n_sessions, Rm, r, work);
    k_scale_vec<<<(R + 255) / 256, 256, 0, st>>>(R, 1.0 / n_sessions, r);
    return (int)hipGetLastError();
}
int tvk_lower_to_upper(hipStream_t st, int n, const double *L, double *U)
{
    k_lower_to_upper<<<(n * n + 255) / 256, 256, 0, st>>>(n, L, U);
    return (int)hipGetLastError();
}
int tvk_batched_matvec(hipStream_t st, int n, int nb, const double *Mx, const double *x, double *y)
{
    if (nb <= 0) return 0;
    k_batched_matvec<<<nb, 256, 0, st>>>(n, Mx, x, y);
    return (int)hipGetLastError();
}
int tvk_vecmat_add(hipStream_t st, int rows, long cols, const double *x, const double *Mx, double *y)
{
    const int blocks = (int)((cols + 255) / 256 > 4096 ? 4096 : (cols + 255) / 256);
    k_vecmat_add<<<blocks, 256, 0, st>>>(rows, cols, x, Mx, y);
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// scoring helpers
// -------------------------------------------------------------------------------------------
// q[j] = sum_i X[i][j] * Y[i][j]  for column-vectors matrices [dim x n] (diag(X^T Y))
__global__ void k_coldot(int dim, long n, long ld, const double *__restrict__ X, const double *__restrict__ Y,
                         double *__restrict__ qv)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < (size_t)n; j += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < dim; ++i) s = __builtin_fma(X[(size_t)i * ld + j], Y[(size_t)i * ld + j], s);
        qv[j] = s;
    }
}
// v[i] = 1 / sqrt(v[i])
__global__ void k_rsqrt_vec(long n, double *__restrict__ v)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) v[e] = 1.0 / sqrt(v[e]);
}
int tvk_rsqrt_vec(hipStream_t st, long n, double *v)
{
    if (n <= 0) return 0;
    k_rsqrt_vec<<<ew_blocks(n), 256, 0, st>>>(n, v);
    return (int)hipGetLastError();
}
// out[i][j] = a[i][j] + b[j][i]   (square n x n)
__global__ void k_add_transpose(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e / n, j = e - i * n;
        out[e] = a[e] + b[(size_t)j * n + i];
    }
}
// out[i][j] = X[i][j] + v[i]  ([dim x n])
__global__ void k_add_colvec(int dim, long n, const double *__restrict__ X, const double *__restrict__ v,
                             double *__restrict__ out)
{
    const size_t tot = (size_t)dim * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x)
        out[e] = X[e] + v[e / n];
}

__global__ void k_axpby(long n, double a, const double *__restrict__ x, double b, const double *__restrict__ y,
                        double *__restrict__ out)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x)
        out[e] = a * x[e] + b * y[e];
}
int tvk_axpby(hipStream_t st, long n, double a, const double *x, double b, const double *y, double *out)
{
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    k_axpby<<<blocks, 256, 0, st>>>(n, a, x, b, y, out);
    return (int)hipGetLastError();
}

int tvk_coldot(hipStream_t st, int dim, long n, const double *X, const double *Y, double *qv, long ld)
{
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    k_coldot<<<blocks, 256, 0, st>>>(dim, n, ld > 0 ? ld : n, X, Y, qv);
    return (int)hipGetLastError();
}
int tvk_add_transpose(hipStream_t st, int n, const double *a, const double *b, double *out)
{
    k_add_transpose<<<(n * n + 255) / 256, 256, 0, st>>>(n, a, b, out);
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// i-vector normalisation (PldaTest::center / rotateLeft / lengthNorm) and classical Gram-Schmidt
// -------------------------------------------------------------------------------------------
__global__ void k_sub_colvec(int dim, long n, const double *__restrict__ X, const double *__restrict__ v,
                             double *__restrict__ out)
{
    const size_t tot = (size_t)dim * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x)
        out[e] = X[e] - v[e / n];
}
// X[:, j] /= sqrt(q[j])
__global__ void k_scale_cols_rsqrt(int dim, long n, double *__restrict__ X, const double *__restrict__ qv)
{
    const size_t tot = (size_t)dim * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x)
        X[e] = X[e] / sqrt(qv[e % n]);
}
int tvk_sub_colvec(hipStream_t st, int dim, long n, const double *X, const double *v, double *out)
{
    if (n <= 0) return 0;
    k_sub_colvec<<<4096, 256, 0, st>>>(dim, n, X, v, out);
    return (int)hipGetLastError();
}
int tvk_scale_cols_rsqrt(hipStream_t st, int dim, long n, double *X, const double *qv)
{
    if (n <= 0) return 0;
    k_scale_cols_rsqrt<<<4096, 256, 0, st>>>(dim, n, X, qv);
    return (int)hipGetLastError();
}

// rv[i] = <Q_i, t>  for i < j   (one workgroup per i)
__global__ __launch_bounds__(256) void k_gs_project(long SV, const double *__restrict__ Q, const double *__restrict__ t,
                                                    double *__restrict__ rv)
{
    __shared__ double red[4];
    const double *qi = Q + (size_t)blockIdx.x * SV;
    double s = 0.0;
    for (long k = threadIdx.x; k < SV; k += 256) s = __builtin_fma(qi[k], t[k], s);
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rv[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// v[k] = t[k] - sum_{i<j} rv[i] Q[i][k] ; partial[b] = sum_k v[k]^2 over this block's k
__global__ __launch_bounds__(256) void k_gs_update(long SV, int j, const double *__restrict__ Q, const double *__restrict__ t,
                                                   const double *__restrict__ rv, double *__restrict__ v,
                                                   double *__restrict__ partial)
{
    __shared__ double red[4];
    double nv = 0.0;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < SV; k += (long)gridDim.x * 256) {
        double a = t[k];
        for (int i = 0; i < j; ++i) a -= rv[i] * Q[(size_t)i * SV + k];
        v[k] = a;
        nv = __builtin_fma(a, a, nv);
    }
    nv = wave_sum_f64(nv);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nv;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// q[k] = norm > 0 ? v[k] / norm : 0, norm = sqrt(sum partial)
__global__ void k_gs_finish(long SV, int nb, const double *__restrict__ v, const double *__restrict__ partial,
                            double *__restrict__ qrow)
{
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += partial[b];
    const double nrm = sqrt(s);
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < SV; k += (long)gridDim.x * blockDim.x)
        qrow[k] = nrm == 0.0 ? 0.0 : v[k] / nrm;
}
// TVAcc::orthonormalizeT: Q (R x SV, output), T (input), scratch: rv[R], v[SV], partial[nb]
int tvk_orthonormalize(hipStream_t st, int R, long SV, const double *Tm, double *Q, double *rv, double *v, double *partial)
{
    const int nb = 512;
    for (int j = 0; j < R; ++j) {
        const double *tj = Tm + (size_t)j * SV;
        if (j > 0) k_gs_project<<<j, 256, 0, st>>>(SV, Q, tj, rv);
        k_gs_update<<<nb, 256, 0, st>>>(SV, j, Q, tj, rv, v, partial);
        k_gs_finish<<<512, 256, 0, st>>>(SV, nb, v, partial, Q + (size_t)j * SV);
    }
    return (int)hipGetLastError();
}
