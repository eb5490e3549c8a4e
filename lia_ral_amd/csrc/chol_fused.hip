// chol_fused.hip -- batched Cholesky of nb SPD matrices of order n (a few hundred) on gfx950, ONE workgroup per
// matrix, left-looking, v_mfma_f64_16x16x4_f64.  Replaces the GEMM-built right-looking factorisation for the
// i-vector systems L_u = I + sum_c N_uc T_c^T S_c^-1 T_c (reference: TVAcc::estimateW / estimateAandC invert
// them one at a time on the host, LIA_SpkTools/src/AccumulateTVStat.cpp:2114-2169, :1702-1795).
//
// Why left-looking: a matrix (1.28 MB at n = 400) does not fit LDS, and a right-looking batch re-reads AND
// re-writes every trailing matrix once per block step.  Here a panel of 32 columns is produced from the
// finished columns to its left in one go,  P = A[j0:, j0:j0+32] - L[j0:, :j0] L[j0:j0+32, :j0]^T,  so the
// matrix is written exactly once and the re-reads (n^3/(6*32) doubles) come out of L2 / MALL.
//
// Workgroup = 8 waves.  The panel is cut in row tiles of 16; a wave owns up to 4 of them and computes each
// tile TRANSPOSED:  D'[c][row] = sum_k L[j0 + c][k] L[row][k]  (MFMA A operand = the 32 panel rows, B operand =
// the tile's own rows).  Both operands are rows of L with k contiguous; of a 32-wide k chunk a lane holds four 16-byte
// pieces (the k order inside a chunk is permuted, identically for A and B: see KOFF_Q / KOFF_V below).  With the A-operand rows
// taken in the order perm(i) = 4 (i & 3) + (i >> 2), lane (i16, q) ends up holding P[row i16][cols 4q .. 4q+3]
// in the 4 result registers: one 32-byte load / store per lane, and -- the point of the transposition -- exactly
// the B-operand layout of the triangular solve  X^T = inv(L_jj) P^T  that follows, again one MFMA chain.
// Wave 0 owns the diagonal block: it factors and inverts it in one sweep over its columns (sweep32: row i of the block
// in lane i, column i of the inverse in lane 32 + i) and publishes both in LDS while the other waves run the k-loops of
// their tiles; it takes off-diagonal tiles last.  k_chol_left2 (the default) stages the panel rows in LDS first and updates
// the diagonal block from there on all waves; k_chol_left is the round-2 form.
#include <atomic>
#include <type_traits>

#include "devutil.h"
#include "lds_attr.h"
#include "tv_kernels.h"

typedef double d2 __attribute__((ext_vector_type(2)));

// k mapping of the row x row dot products: of a 32-wide chunk starting at k, lane quarter q holds the eight columns
// k + KOFF_V v + KOFF_Q q + {0, 1}, v = 0..3, for BOTH operands (any bijection works as long as they agree).  With (2, 8) the four
// quarters of a row read 64 contiguous bytes per load instruction, so a 128-byte line is fetched by two instructions; the round-1..3
// mapping (8, 2) gave every lane 64 contiguous bytes of its own, every line was touched by FOUR instructions 16 bytes at a time, and
// the k-loops -- which wait for these loads, not for the MFMAs -- thrashed the 32 KB L1 as soon as more chunks were in flight.
#ifndef KOFF_Q
#define KOFF_Q 2
#define KOFF_V 8
#endif
// The LDS copy of the panel rows keeps the round-1..3 placement, which is free of bank conflicts for the 16-lane groups of
// ds_read_b128 (row stride = an odd number of 16-byte units, lane quarter q 64 bytes on): stage_panel stores the 16-byte units of
// every 32-column chunk TRANSPOSED as a 4 x 4 block (logical unit 4 v + q -> physical unit 4 q + v), so lane quarter q finds its
// columns 8 v + 2 q, + 1 at physical offset 8 q + 2 v.  (With the (2, 8) offsets on the LDS side too: SQ_LDS_BANK_CONFLICT 8.5 M ->
// 32 M cycles of 92 M LDS cycles in k_chol_left2, 0 -> 20.6 M of 46 M in k_uut.)
#if KOFF_Q == 2
#define LOFF_Q 8
#define LOFF_V 2
#define PAN_UNIT(sg) (((sg) & ~15) | (((sg) & 3) << 2) | (((sg) >> 2) & 3))
#else
#define LOFF_Q KOFF_Q
#define LOFF_V KOFF_V
#define PAN_UNIT(sg) (sg)
#endif

// CHOL_ABL (compile-time, timing experiments only -- tools/chol_ablate.sh; results are wrong when != 0), k_chol_left: 1 = no factorisation /
// inversion sweep of the diagonal block, 2 = no triangular solve + stores of the tiles, 4 = no k-loops of the off-diagonal tiles,
// 8 = no k-loop of the diagonal block's update, 16 = triangular solve kept but no stores of the tiles, 32 = the barrier that ends a panel does
// not wait for the stores (s_barrier alone)
#ifndef CHOL_ABL
#define CHOL_ABL 0
#endif

// CHOL_PROF (compile-time, tools/chol_probe.sh): waves 0 and 1 of workgroup 0 leave s_memtime stamps at the phase boundaries of
// every panel in a device array the probe prints (kernel 0 = k_chol_left, 1 = k_trinv_left, 2 = k_uut; [kernel][wave][panel][16])
#ifndef CHOL_PROF
#define CHOL_PROF 0
#endif
long long *g_chol_prof = nullptr; // device address of the stamp array once a -DCHOL_PROF build has launched a kernel
#if CHOL_PROF
__device__ long long g_chol_prof_dev[3 * 2 * 32 * 16];
#define PROF_INIT(kid, wv, ln) long long *prof_ = (blockIdx.x == 0 && (wv) < 2 && (ln) == 0) ? g_chol_prof_dev + ((kid) * 2 + (wv)) * 32 * 16 : nullptr; int prof_i_ = 0
#define PROF_PANEL(p) do { if (prof_) { prof_ = prof_ - (prof_ - g_chol_prof_dev) % (32 * 16) + ((p) & 31) * 16; prof_i_ = 0; } } while (0)
#define STAMP() do { if (prof_ && prof_i_ < 16) prof_[prof_i_++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
static void prof_host_init() { if (!g_chol_prof) { void *p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_chol_prof_dev)) == hipSuccess) { g_chol_prof = (long long *)p; (void)hipMemset(p, 0, sizeof(long long) * 3 * 2 * 32 * 16); } } }
#else
#define PROF_INIT(kid, wv, ln) do { } while (0)
#define PROF_PANEL(p) do { } while (0)
#define STAMP() do { } while (0)
static void prof_host_init() {}
#endif

// A/B switch of the calling host thread: 0 = every wave fetches the panel rows itself (the round-1 kernels)
#define g_chol_lds (gmmiv_kopts_cur().chol_lds) // option of the calling context (ctx.h: gmmiv_kopts)

namespace {

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define PIN_V(x) asm volatile("" : "+v"(x)) // orders the VALU chain producing x against later loads / asm statements
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// TW = row tiles per wave and pass (2 x (2 + TW) x 16 operand VGPRs in flight + TW x 16 accumulators) is a template parameter of
// the kernels, deduced by the helpers below from the arrays they are handed.  Round 2 ran TW = 3 everywhere; measured per 1024
// systems of order 400 (tools/chol_probe.hip): k_chol_left2 1.31 / 1.21 / 1.11 ms for TW = 3 / 2 / 1, k_trinv_left 0.85 / 0.82 /
// 0.80, k_uut 1.39 / 1.12 / 1.17, TW = 4: 1.67 / 0.96 / 1.49 -- these kernels are bound by latency, not by operand reuse: fewer
// tiles per pass mean more, shorter passes that balance better over the waves, and no spilled accumulators.

// One 32-wide k chunk of operands: the 32 panel rows (a0 / a1: rows perm(i16), 16 + perm(i16)) and CNT row tiles,
// 64 contiguous bytes per lane and row.
template <int CNT> struct RowOps { d2 a0[4], a1[4], b[CNT][4]; };

template <int CNT, int TWT>
__device__ __forceinline__ void rows_load(RowOps<CNT> &o, const double *pa0, const double *pa1, const double *(&pb)[TWT], int k)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        o.a0[v] = *(const d2 *)(pa0 + k + KOFF_V * v);
        o.a1[v] = *(const d2 *)(pa1 + k + KOFF_V * v);
    }
#pragma unroll
    for (int u = 0; u < CNT; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) o.b[u][v] = *(const d2 *)(pb[u] + k + KOFF_V * v);
}
template <int CNT, bool NEG, int TWT>
__device__ __forceinline__ void rows_mfma(const RowOps<CNT> &o, d4 (&acc)[TWT][2])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double x0 = NEG ? -o.a0[e >> 1][e & 1] : o.a0[e >> 1][e & 1];
        const double x1 = NEG ? -o.a1[e >> 1][e & 1] : o.a1[e >> 1][e & 1];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            acc[u][0] = MFMA_F64(x0, o.b[u][e >> 1][e & 1], acc[u][0]);
            acc[u][1] = MFMA_F64(x1, o.b[u][e >> 1][e & 1], acc[u][1]);
        }
    }
}
// keeps the operand loads of the NEXT chunk ahead of the MFMAs of the current one: the memory clobber pins them in
// the IR (hipcc otherwise sinks them to their first use), the scheduling barrier in the machine scheduler
#define ROWS_FENCE()                         \
    do {                                     \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// acc[u][ct] (+/-)= sum over k in [kb, ke) of panel_row[k] * tile_row[k]   ((ke - kb) % 32 == 0), register
// double-buffered one chunk ahead.  Every load is unconditional (the last prefetch is clamped to the final chunk and
// thrown away): with a conditional prefetch the compiler merges "issued" and "not issued" at the join and waits
// vmcnt(0), i.e. for the prefetch itself.
template <int CNT, bool NEG, int TWT>
__device__ __forceinline__ void rowdot(const double *pa0, const double *pa1, const double *(&pb)[TWT], int kb, int ke,
                                       d4 (&acc)[TWT][2])
{
    if (ke - kb < 32) return;
    RowOps<CNT> A, B;
    const int last = ke - 32;
    rows_load<CNT>(A, pa0, pa1, pb, kb);
    int k = kb;
    for (; k + 64 <= ke; k += 64) {
        rows_load<CNT>(B, pa0, pa1, pb, k + 32);
        ROWS_FENCE();
        rows_mfma<CNT, NEG>(A, acc);
        ROWS_FENCE();
        rows_load<CNT>(A, pa0, pa1, pb, k + 64 < last ? k + 64 : last);
        ROWS_FENCE();
        rows_mfma<CNT, NEG>(B, acc);
        ROWS_FENCE();
    }
    if (k < ke) rows_mfma<CNT, NEG>(A, acc);
}
template <bool NEG, int TWT>
__device__ __forceinline__ void rowdot_n(int cnt, const double *pa0, const double *pa1, const double *(&pb)[TWT], int kb,
                                         int ke, d4 (&acc)[TWT][2])
{
    switch (cnt) { // wave-uniform
    case 1: rowdot<1, NEG>(pa0, pa1, pb, kb, ke, acc); break;
    case 2: rowdot<(TWT >= 2 ? 2 : TWT), NEG>(pa0, pa1, pb, kb, ke, acc); break;
    case 3: rowdot<(TWT >= 3 ? 3 : TWT), NEG>(pa0, pa1, pb, kb, ke, acc); break;
    case 4: rowdot<(TWT >= 4 ? 4 : TWT), NEG>(pa0, pa1, pb, kb, ke, acc); break;
    default: break;
    }
}

// ---- panel rows from LDS --------------------------------------------------------------------------------------------------
// The 32 panel rows are the A operand of EVERY wave's MFMAs: fetched by each wave itself they were 40 % of the kernel's L2 / MALL
// traffic (8 KB per wave and 32-column chunk next to 12 KB of its own tile rows) -- and these kernels are bound by exactly that
// traffic (2.8 GB per launch of 256 matrices).  The panel block [32 rows x klen columns] (at most 32 x n doubles = 103 KB at
// n = 400) is staged ONCE per panel by the whole workgroup; a wave then reads its A operands with 8 ds_read_b128 per chunk.
// Row stride S (doubles) with S / 2 odd: the 16 lanes of a b128 read (16 different rows, same column) hit 16 distinct 4-bank groups.
__device__ __forceinline__ int pan_stride(int n) { return (n & 2) ? n : n + 2; }   // n even: S / 2 odd
// rows row0 .. row0 + 31 (clamped to n - 1), columns [kbase, kbase + klen) of the row-major matrix X -> pan[row][0 .. klen).
// Thread t serves row t >> 4 and the 16-byte segments (t & 15) + 16 i of it (16 threads = 256 contiguous bytes per step).  The
// loads of a batch are ALL issued before the first LDS store: written as "load, store, next" hipcc waits vmcnt(0) in every trip
// -- one full L2 / MALL latency per 16-byte segment and thread, up to 13 in a row at n = 400, 10 us per panel: that loop, not
// the barriers, was the 37 % of k_chol_left that remained with all arithmetic compiled out (round 2's ablation).  Loads past
// the row's end are clamped to its last segment (never stored).
template <int NB, int TPR>
__device__ __forceinline__ void stage_batch(double *prow, const double *xrow, int segs, int s0)
{
    d2 v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        int sg = s0 + TPR * u;
        sg = sg < segs ? sg : segs - 1;
        v[u] = *(const d2 *)(xrow + 2 * sg);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
        if (s0 + TPR * u < segs) *(d2 *)(prow + 2 * PAN_UNIT(s0 + TPR * u)) = v[u];
}
template <int TPR = 16> // threads per panel row: 16 for a workgroup of 512, 32 for 1024
__device__ __forceinline__ void stage_panel(double *pan, int S, const double *X, long n, long row0, int kbase, int klen, int tid)
{
    const int segs = klen >> 1; // 16-byte segments per row
    if (segs <= 0) return;
    const int row = tid / TPR;
    long r = row0 + row;
    r = r < n ? r : n - 1;
    const double *xrow = X + r * n + kbase;
    double *prow = pan + row * S;
    int s0 = tid % TPR;
    for (; s0 + TPR * 7 < segs; s0 += TPR * 8) stage_batch<8, TPR>(prow, xrow, segs, s0);   // full batches of 8 segments per thread
    if (s0 < segs) {
        if (s0 + TPR * 3 < segs) stage_batch<8, TPR>(prow, xrow, segs, s0);
        else stage_batch<4, TPR>(prow, xrow, segs, s0);
    }
}
template <int CNT> struct TileOps { d2 b[CNT][4]; };
struct PanOps { d2 a0[4], a1[4]; };
template <int CNT, int TWT>
__device__ __forceinline__ void tiles_load(TileOps<CNT> &o, const double *(&pb)[TWT], int k)
{
#pragma unroll
    for (int u = 0; u < CNT; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) o.b[u][v] = *(const d2 *)(pb[u] + k + KOFF_V * v);
}
__device__ __forceinline__ void pan_load(PanOps &o, const double *la0, const double *la1, int kl)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        o.a0[v] = *(const d2 *)(la0 + kl + LOFF_V * v);
        o.a1[v] = *(const d2 *)(la1 + kl + LOFF_V * v);
    }
}
template <int CNT, bool NEG, int TWT>
__device__ __forceinline__ void pan_mfma(const PanOps &a, const TileOps<CNT> &o, d4 (&acc)[TWT][2])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double x0 = NEG ? -a.a0[e >> 1][e & 1] : a.a0[e >> 1][e & 1];
        const double x1 = NEG ? -a.a1[e >> 1][e & 1] : a.a1[e >> 1][e & 1];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            acc[u][0] = MFMA_F64(x0, o.b[u][e >> 1][e & 1], acc[u][0]);
            acc[u][1] = MFMA_F64(x1, o.b[u][e >> 1][e & 1], acc[u][1]);
        }
    }
}
// rowdot with the panel rows in LDS (la0 / la1: the lane's two panel rows + 8 q, LDS column = k - kbase); tile rows still come
// from memory, register double-buffered one chunk ahead; the LDS operands of the next chunk are requested before the MFMAs of
// the current one.
template <int CNT, bool NEG, int TWT>
__device__ __forceinline__ void rowdot_lds(const double *la0, const double *la1, const double *(&pb)[TWT], int kb, int ke, int kbase,
                                           d4 (&acc)[TWT][2])
{
    // (A third rotating operand buffer -- tile rows requested two chunks ahead -- was tried: 120-137 spilled VGPRs in all three
    // kernels and the gain of the LDS panel gone.  Two buffers it is.)
    if (ke - kb < 32) return;
    TileOps<CNT> A, B;
    PanOps xa, xb;
    const int last = ke - 32;
    tiles_load<CNT>(A, pb, kb);
    pan_load(xa, la0, la1, kb - kbase);
    int k = kb;
    for (; k + 64 <= ke; k += 64) {
        tiles_load<CNT>(B, pb, k + 32);
        pan_load(xb, la0, la1, k + 32 - kbase);
        ROWS_FENCE();
        pan_mfma<CNT, NEG>(xa, A, acc);
        ROWS_FENCE();
        const int kn = k + 64 < last ? k + 64 : last;
        tiles_load<CNT>(A, pb, kn);
        pan_load(xa, la0, la1, kn - kbase);
        ROWS_FENCE();
        pan_mfma<CNT, NEG>(xb, B, acc);
        ROWS_FENCE();
    }
    if (k < ke) pan_mfma<CNT, NEG>(xa, A, acc);
}
// (Round 3 tried to request the tile rows further ahead, on the theory that the k-loops wait for memory LATENCY -- ~3 k cycles per
// chunk next to 0.5 k of MFMAs in the stamps: (a) four rotating 32-wide buffers refilled behind their MFMAs, tail steps guarded by
// wave-uniform ifs -- hipcc answers the guarded blocks with s_waitcnt vmcnt(0) at the loop head: k_chol_left2 0.97 -> 1.32 ms,
// k_trinv_left 0.78 -> 0.97; (b) 64-wide tile-row chunks in two buffers, LDS operands on 32-wide steps, precise waits: 0.97 -> 1.20
// (spills) and 0.78 -> 0.82 (none); (c) k_chol_left2 alone (its tile rows come from L2: 1.2 TB/s of fetches against 4.6 for
// k_trinv_left), three rotating 32-wide buffers, a main loop without conditionals, LDS operands on half steps, no spills, exact
// vmcnt waits: 0.962 -> 0.984.  More requests in flight never helped: seven waves x 4 KB per chunk per ~3.2 k cycles x 256 CUs are
// 5.5 TB/s of tile rows that no other wave shares -- the loops are bound by the THROUGHPUT of the memory path (L2 / fabric / HBM,
// 64-byte pieces of 16 rows per instruction), not by its latency.  What would help is fewer passes over L (Cholesky
// family), not deeper queues.)
// dispatcher over the tile count; the operand source is a template parameter of the kernels (two code paths in one kernel cost
// 150 spilled VGPRs)
template <bool LDS, bool NEG, int TWT>
__device__ __forceinline__ void rowdot_sel(int cnt, const double *pa0, const double *pa1, const double *la0, const double *la1,
                                           const double *(&pb)[TWT], int kb, int ke, int kbase, d4 (&acc)[TWT][2])
{
    if (!LDS) { rowdot_n<NEG>(cnt, pa0, pa1, pb, kb, ke, acc); return; }
    switch (cnt) {
    case 1: rowdot_lds<1, NEG>(la0, la1, pb, kb, ke, kbase, acc); break;
    case 2: rowdot_lds<(TWT >= 2 ? 2 : TWT), NEG>(la0, la1, pb, kb, ke, kbase, acc); break;
    case 3: rowdot_lds<(TWT >= 3 ? 3 : TWT), NEG>(la0, la1, pb, kb, ke, kbase, acc); break;
    case 4: rowdot_lds<(TWT >= 4 ? 4 : TWT), NEG>(la0, la1, pb, kb, ke, kbase, acc); break;
    default: break;
    }
}

// X^T = inv(L_jj) P^T for one row tile: P in acc[0..1] (lane (i16, q): P[row i16][16 ct + 4 q + r]), the 12 operand
// values of inv(L_jj) in la0 / la1; result x0 (columns 4q..4q+3) and x1 (columns 16 + 4q ..) of the same rows
struct LinvOps { double la0[4], la1[2][4]; };
__device__ __forceinline__ void linv_ops_load(LinvOps &o, const double (*linv)[34], int perm, int q)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o.la0[r] = linv[perm][4 * q + r];
        o.la1[0][r] = linv[16 + perm][4 * q + r];
        o.la1[1][r] = linv[16 + perm][16 + 4 * q + r];
    }
}
__device__ __forceinline__ void tile_trsm(const LinvOps &o, const d4 (&p)[2], d4 &x0, d4 &x1)
{
    x0 = d4{0.0, 0.0, 0.0, 0.0};
    x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x0 = MFMA_F64(o.la0[r], p[0][r], x0);
        x1 = MFMA_F64(o.la1[0][r], p[0][r], x1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) x1 = MFMA_F64(o.la1[1][r], p[1][r], x1);
}

} // namespace

// the factored diagonal block and its inverse leave LDS: all 512 threads, two elements each
__device__ __forceinline__ void chol_block_out(const double (*pj)[34], const double (*linv)[34], double *Lm, double *iv, long n,
                                               int j0, int w, int tid)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = tid + 512 * t, i = e >> 5, k = e & 31;
        const bool in = i < w && k < w;
        const double vl = pj[i][k], vi = linv[i][k];
        if (in) Lm[(long)(j0 + i) * n + j0 + k] = vl;
        iv[(size_t)(j0 >> 5) * 1024 + e] = in ? vi : 0.0;
    }
}

// A[r][c0 .. c0 + 3] of the matrix being factored: from the row-major array itself (factorisation in place) or from
// PACKED lower rows (r (r + 1) / 2 + c, the layout the L = N TETt GEMM produces) with diag_add on the diagonal -- the
// unpack kernel (a 2 x 1.3 GB round trip per 1024 systems of order 400) disappears.  Entries above the diagonal come out
// as whatever follows in the packed array; the factorisation never uses them.
template <bool PK> // PK: the source is the packed lower-row input (Apk), else the full matrix itself
__device__ __forceinline__ d4 chol_src4t(const double *Lm, const double *Apk, long n, long r, long c0, double diag_add)
{
    if (!PK) {
        const double *p = Lm + r * n + c0;
        const d2 lo = *(const d2 *)p, hi = *(const d2 *)(p + 2);
        return d4{lo[0], lo[1], hi[0], hi[1]};
    }
    const double *p = Apk + r * (r + 1) / 2 + c0;
    d4 v = {p[0], p[1], p[2], p[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (c0 + k == r) ? diag_add : 0.0;
    return v;
}
__device__ __forceinline__ d4 chol_src4(const double *Lm, const double *Apk, long n, long r, long c0, double diag_add)
{
    return Apk ? chol_src4t<true>(Lm, Apk, n, r, c0, diag_add) : chol_src4t<false>(Lm, Apk, n, r, c0, diag_add);
}

// Afull[b]: n x n row-major, lower triangle read, overwritten by the factor (diagonal blocks: upper part zeroed;
// elsewhere the upper triangle is left as it was).  invd[b][kb][32][32]: inverse of the kb-th 32 x 32 diagonal
// block of the factor (zero padded).  status[b] = 1 on a non-positive pivot.  n must be even (16-byte rows).
// Factorisation and inversion of the 32 x 32 diagonal block by ONE wave in ONE sweep over its columns, one FMA stream for both:
// lanes 0..31 hold row i = lane of the (updated, symmetric) block, lanes 32..63 the running sums of column i = lane - 32 of
// X = L^-1 (v[] either way; a[] is read on lanes < 32 only).  Results: blk[0] = pj[i][j] = L[i][j] (zeros above the diagonal),
// blk[1] = linv[j][i] = X[j][i]; bad = 1 when a pivot is not a positive normal number (the pivot is replaced by 1).
//
// This sweep is the longest serial piece of a panel (round 3's in-kernel stamps: 43 % of k_chol_left2 at order 400, ~680 cycles
// per column), so what sits on the column-to-column dependence is kept minimal (CHOL_SWEEP 1):
//   * the recurrence runs on UNSCALED columns (the L D L^T form): v[k] += coef c[k] with coef = -v[j] / d_j needs 1 / d_j only
//     (v_rcp_f64 + two Newton steps = 5 dependent instructions) -- sqrt(d_j) and 1 / sqrt(d_j) (17 dependent instructions) scale
//     the OUTPUT and are evaluated once per block after the loop, by lane j for its own pivot, not once per column on the chain;
//   * the pivot d_j and the next column's entry c[j + 1] come out of the registers of lanes j and j + 1 with v_readlane (4 per
//     column: no SGPR spilling), so the LDS round trip of a column (write by 32 lanes, broadcast reads) is off the chain: it only
//     feeds the 30 - j updates that nothing waits for;
//   * no exec-masked branches (both halves store through one per-lane address): the 32 columns are one scheduling region.
// CHOL_SWEEP 0 = the round-1..3 form (column through LDS, rsqrt chain per column), kept for A/B runs of tools/chol_probe.
#ifndef CHOL_SWEEP
#define CHOL_SWEEP 1
#endif
__device__ __forceinline__ void rsqrt_sqrt(double dj, double &rs, double &sq)
{
    // 1/sqrt(d) and sqrt(d) from v_rsq_f64 + Newton / Goldschmidt steps (a dozen dependent FMAs instead of the ~60 instructions
    // of an IEEE sqrt followed by an IEEE divide)
    rs = __builtin_amdgcn_rsq(dj);
    const double hd = 0.5 * dj;
    rs = rs * __builtin_fma(-hd * rs, rs, 1.5);
    rs = rs * __builtin_fma(-hd * rs, rs, 1.5);
    sq = dj * rs;
    sq = __builtin_fma(__builtin_fma(-sq, sq, dj), 0.5 * rs, sq);
    rs = __builtin_fma(__builtin_fma(-sq, rs, 1.0), rs, rs);
}
// one column of sweep32 (CHOL_SWEEP 1); J is a template parameter so that every loop bound is a constant for the front end.
// c[] holds column J - 1 on entry (entries k >= J + 1) and column J on exit (k >= J + 2): every entry is reloaded right behind the
// update that consumed it (one buffer: two would cost 60 more VGPRs next to v[], and the kernel is at its cap of 256), a whole
// step before it is needed.  A scheduling barrier behind every chunk keeps the loads where they are written (hipcc otherwise sinks
// them to their first use, and the LDS latency lands on the chain).
template <int J>
__device__ __forceinline__ void sweep_column(double (&v)[32], double (&c)[32], double &cprev, int lane, double (*col)[64], double *piv,
                                             int &bad)
{
    // v[J] is final: pivot and the next column's entry from the registers of lanes J, J + 1
    const int ph = __builtin_amdgcn_readlane(__double2hiint(v[J]), J), pl = __builtin_amdgcn_readlane(__double2loint(v[J]), J);
    const bool okp = ph > 0 && ph < 0x7ff00000; // positive, normal, finite (scalar test on the high word)
    if (!okp) bad = 1;
    const double dj = okp ? __hiloint2double(ph, pl) : 1.0;
    piv[J] = dj; // uniform store
    const double *cj = col[J & 1];
    // the updates of step J - 1 (k = J + 1 .. 31, the next pivot's column first) in six chunks under the dependent chain of 1 / d_J
    constexpr int NF = J > 0 ? 31 - J : 0, CS = (NF + 5) / 6;
#define DCHUNK(ci)                                                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < CS; ++kk)                                                                                \
    {                                                                                                                                \
        const int k = J + 1 + (ci) * CS + kk;                                                                                        \
        if (k < 32) {                                                                                                                \
            v[k] = __builtin_fma(cprev, c[k], v[k]);                                                                                 \
            PIN_V(v[k]);                                                                                                             \
            if (k >= J + 2) c[k] = cj[k];                                                                                            \
        }                                                                                                                            \
    }                                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    if (J == 0) {
#pragma unroll
        for (int k = 2; k < 32; ++k) c[k] = cj[k];
    }
    double r = __builtin_amdgcn_rcp(dj);
    PIN_V(r);
    DCHUNK(0)
    double e = __builtin_fma(-dj, r, 1.0);
    PIN_V(e);
    DCHUNK(1)
    r = __builtin_fma(e, r, r);
    PIN_V(r);
    DCHUNK(2)
    e = __builtin_fma(-dj, r, 1.0);
    PIN_V(e);
    DCHUNK(3)
    r = __builtin_fma(e, r, r);
    PIN_V(r);
    DCHUNK(4)
    const double coef = -v[J] * r;
    DCHUNK(5)
#undef DCHUNK
    if (J + 1 < 32) {
        const double cn = readlane_f64(v[J], J + 1);
        v[(J + 1) & 31] = __builtin_fma(coef, cn, v[(J + 1) & 31]);
        PIN_V(v[(J + 1) & 31]);
        col[(J + 1) & 1][lane] = v[(J + 1) & 31];
        wave_sync(); // compiler-level only: LDS is in order per wave
    }
    cprev = coef;
}
template <int J>
__device__ __forceinline__ void sweep_columns(double (&v)[32], double (&c)[32], double &cprev, int lane, double (*col)[64], double *piv,
                                              int &bad)
{
    if constexpr (J < ((CHOL_ABL & 1) ? 1 : 32)) {
        sweep_column<J>(v, c, cprev, lane, col, piv, bad);
        sweep_columns<J + 1>(v, c, cprev, lane, col, piv, bad);
    }
}
__device__ __forceinline__ void sweep32(const double (&a)[32], int lane, double (*col)[64], double *piv, double (*blk)[32][34], int &bad)
{
    const bool lo = lane < 32;
    const int li = lane & 31;
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = lo ? a[k] : 0.0;
    col[0][lane] = v[0]; // lanes >= 32 write the unused upper half: no branch
#if CHOL_SWEEP
    // lanes >= 32 run the SAME recurrence on u = -v started from the unit vector e_i (u[k] += (-u[j] / d_j) c[k]): X[j][i] = u[j] / sqrt(d_j)
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = lo ? v[k] : (k == li ? 1.0 : 0.0);
    double c[32];       // the column the pending updates use
    double cprev = 0.0; // coef of step j - 1
    wave_sync();
    sweep_columns<0>(v, c, cprev, lane, col, piv, bad);
    // (the updates of step 31 do not exist: k >= 33)
    // output: column j of both results is the unscaled column times 1 / sqrt(d_j), evaluated once per block by lanes j / 32 + j
    wave_sync();
    double rs, sq;
    rsqrt_sqrt(piv[li], rs, sq);
    col[0][lane] = rs;
    wave_sync();
    double *out = &blk[0][0][0] + (lo ? li * 34 : 32 * 34 + li);
    const int step = lo ? 1 : 34;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double y = v[j] * col[0][j];
        out[j * step] = lane == j ? sq : (lane < j ? 0.0 : y); // lanes < 32: zeros above the diagonal, sqrt(d_j) on it
    }
#else
#pragma unroll
    for (int j = 0; j < ((CHOL_ABL & 1) ? 1 : 32); ++j) {
        const double *cj = col[j & 1];
        wave_sync();
        double c[32];
#pragma unroll
        for (int k = j; k < 32; ++k) c[k] = cj[k];
        double dj = c[j];
        if (!(dj > 0.0)) { bad = 1; dj = 1.0; }
        double rs, sq;
        rsqrt_sqrt(dj, rs, sq);
        const double y = v[j] * rs;
        const double xj = (j == li) ? rs : (j > li ? -y : 0.0);
        const double coef = (lo ? -y : xj) * rs;
        if (j + 1 < 32) {
            v[j + 1] = __builtin_fma(coef, c[j + 1], v[j + 1]);
            PIN_V(v[j + 1]);
            if (lo) col[(j + 1) & 1][li] = v[j + 1];
        }
        if (lo) blk[0][li][j] = (li == j) ? sq : (li > j ? y : 0.0);
        else blk[1][j][li] = xj;
#pragma unroll
        for (int k = j + 2; k < 32; ++k) {
            v[k] = __builtin_fma(coef, c[k], v[k]);
            PIN_V(v[k]);
        }
    }
#endif
    wave_sync();
}

template <bool use_lds>
__global__ __launch_bounds__(512, 1) void k_chol_left(int n_, double *Afull, double *invd, long sinv, int *status, const double *Apacked,
                                                      long spk, double diag_add)
{
    constexpr int TW = 3;
    __shared__ __attribute__((aligned(16))) double blk[2][32][34]; // the factored diagonal block and its inverse (sweep32)
    double (*pj)[34] = blk[0], (*linv)[34] = blk[1];
    __shared__ __attribute__((aligned(16))) double col[2][64], piv[32];
    // dynamic: the partial updates of the diagonal block, one per wave (slab[8][32][33]) -- and, once wave 0 has taken them into its
    // registers, the panel rows L[j0 .. j0 + 31][0 .. j0) for the off-diagonal tiles (use_lds; the two never live at the same time)
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double (*slab)[32][33] = (double (*)[32][33])dyn_lds;
    double *pan = dyn_lds;
    const int S = pan_stride(n_);
    const long n = n_;
    double *Lm = Afull + (size_t)blockIdx.x * n * n;
    const double *Apk = Apacked ? Apacked + (size_t)blockIdx.x * spk : nullptr; // input, when it is not Afull itself
    double *iv = invd + (size_t)blockIdx.x * sinv;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    // tiles go to waves 1 2 3 5 6 7 4 0 in turn: wave 0 owns the serial diagonal work, wave 4 shares its SIMD
    const int slot = wave == 0 ? 7 : (wave == 4 ? 6 : (wave < 4 ? wave - 1 : wave - 2));
    int bad = 0;
    d4 acc[TW][2];
    PROF_INIT(0, wave, lane0);

    for (int j0 = 0; j0 < n_; j0 += 32) {
        PROF_PANEL(j0 >> 5);
        STAMP(); // 0: panel start
        // lane-derived indices are recomputed per panel: laundered, so that the compiler does not hoist every address
        // expression of the panel body out of the loop and keep (spill) it for the whole kernel
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int i16 = lane & 15, q = lane >> 4, perm = 4 * (i16 & 3) + (i16 >> 2);
        const int w = (n_ - j0) < 32 ? (n_ - j0) : 32;
        const int below = n_ - j0 - 32;                       // rows under the diagonal block
        const int nt = below > 0 ? (below + 15) / 16 : 0;     // off-diagonal row tiles
        const int mine = nt > slot ? (nt - slot + 7) / 8 : 0; // tiles of this wave: slot, slot + 8, ...
        const int ngroups = (((nt + 7) / 8) + TW - 1) / TW; // uniform over the workgroup
        long ra0 = j0 + perm, ra1 = j0 + 16 + perm;
        ra0 = ra0 < n ? ra0 : n - 1;
        ra1 = ra1 < n ? ra1 : n - 1;
        const double *pa0 = Lm + ra0 * n + KOFF_Q * q, *pa1 = Lm + ra1 * n + KOFF_Q * q;

        // ---------------- diagonal block ----------------
        // step 1, all waves: the update sum_{k < j0} L[j0 + i][k] L[j0 + c][k] of the 32 x 32 block is cut in 8 k ranges
        // (one wave alone would walk the whole range at one memory latency per 32 columns and hold up everybody at the
        // barrier below); the partial blocks meet in LDS.  Wave 0's partial starts from the block of A itself.
        {
            const double *pb[TW];
            long rr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                long r = j0 + 16 * u + i16;
                rr[u] = r < n ? r : n - 1;
                pb[u] = Lm + rr[u] * n + KOFF_Q * q;
                acc[u][0] = acc[u][1] = d4{0.0, 0.0, 0.0, 0.0};
            }
            pb[2] = pb[0];
            if (wave == 0) {
                if (w == 32) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) acc[u][ct] = chol_src4(Lm, Apk, n, rr[u], j0 + 16 * ct + 4 * q, diag_add);
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * u + i16, k = 16 * ct + 4 * q + r;
                                double v = 0.0;
                                if (i < w && k < w) {
                                    if (!Apk) v = Lm[(long)(j0 + i) * n + j0 + k];
                                    else if (k <= i) v = Apk[(long)(j0 + i) * (j0 + i + 1) / 2 + j0 + k] + (k == i ? diag_add : 0.0);
                                }
                                acc[u][ct][r] = v;
                            }
                }
            }
            const int p32 = j0 >> 5;
            if (!(CHOL_ABL & 8)) rowdot<2, true>(pa0, pa1, pb, 32 * ((wave * p32) >> 3), 32 * (((wave + 1) * p32) >> 3), acc);
            const double unit = wave == 0 ? 1.0 : 0.0; // rows / columns beyond n: identity
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * u + i16, k = 16 * ct + 4 * q + r;
                        slab[wave][i][k] = (i < w && k < w) ? acc[u][ct][r] : (i == k ? unit : 0.0);
                    }
        }
        STAMP(); // 1: diagonal update done, partial written
        __syncthreads(); // the 8 partial updates are in LDS
        STAMP(); // 2
        // step 2, wave 0: row i = lane & 31 of the block in registers, factor, invert
        const int li = lane & 31;
        if (use_lds && wave == 0) { // the summed block moves to pj: the slab memory is about to become the panel buffer
            double a[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = slab[0][li][k];
#pragma unroll
            for (int sl = 1; sl < 8; ++sl) {
                double t[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) t[k] = slab[sl][li][k];
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    a[k] += t[k];
                    PIN_V(a[k]);
                }
            }
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 32; ++k) pj[li][k] = a[k];
            }
        }
        STAMP(); // 3: (wave 0) slabs summed into pj
        if (use_lds && j0 > 0) {
            __syncthreads(); // wave 0 has taken the block: the slab memory is free
            stage_panel(pan, S, Lm, n, j0, 0, j0, tid);
            __syncthreads(); // the panel rows are in LDS for everybody (wave 0 included: it takes its tiles last)
        }
        STAMP(); // 4: panel staged
        if (wave == 0) {
            double a[32];
            if (use_lds) {
                wave_sync();
#pragma unroll
                for (int k = 0; k < 32; ++k) a[k] = pj[li][k];
                wave_sync(); // every lane holds its row before the sweep starts overwriting pj
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) a[k] = slab[0][li][k];
#pragma unroll
                for (int sl = 1; sl < 8; ++sl) {
                    double t[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) t[k] = slab[sl][li][k];
#pragma unroll
                    for (int k = 0; k < 32; ++k) {
                        a[k] += t[k];
                        PIN_V(a[k]);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(3); // the serial part: do not queue behind the MFMA stream of the wave sharing this SIMD
            sweep32(a, lane, col, piv, blk, bad); // factor + invert, results in pj / linv
            __builtin_amdgcn_s_setprio(0);
        }

        STAMP(); // 5: (wave 0) block factored and inverted
        // ---------------- off-diagonal tiles, TW per wave and group ----------------
        for (int g = 0; g < ngroups; ++g) {
            int cnt = mine - TW * g;
            cnt = cnt < 0 ? 0 : (cnt > TW ? TW : cnt);
            const double *pb[TW];
            long rows[TW];
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                long r = j0 + 32 + 16L * (slot + 8 * (TW * g + u)) + i16;
                rows[u] = r;
                r = r < n ? r : n - 1;
                pb[u] = Lm + r * n + KOFF_Q * q;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if (u < cnt) acc[u][ct] = (CHOL_ABL & 64) ? d4{0.0, 0.0, 0.0, 0.0} : chol_src4(Lm, Apk, n, r, j0 + 16 * ct + 4 * q, diag_add);
                }
            }
            if (!(CHOL_ABL & 4)) rowdot_sel<use_lds, true>(cnt, pa0, pa1, pan + perm * S + LOFF_Q * q, pan + (16 + perm) * S + LOFF_Q * q, pb, 0, j0, 0, acc);
            if (g == 0) {
                STAMP(); // 6: k-loops of the first tile group done
                __syncthreads(); // inv(L_jj) and L_jj are in LDS
                STAMP(); // 7
                chol_block_out(pj, linv, Lm, iv, n, j0, w, tid);
            }
            if (cnt > 0 && !(CHOL_ABL & 2)) {
                LinvOps lo;
                linv_ops_load(lo, linv, perm, q);
#pragma unroll
                for (int u = 0; u < TW; ++u) {
                    if (u < cnt) {
                        d4 x0, x1;
                        tile_trsm(lo, acc[u], x0, x1);
                        if (CHOL_ABL & 16) { asm volatile("" ::"v"(x0), "v"(x1)); continue; }
                        if (rows[u] < n) {
                            double *p = Lm + rows[u] * n + j0 + 4 * q;
                            *(d2 *)p = d2{x0[0], x0[1]};
                            *(d2 *)(p + 2) = d2{x0[2], x0[3]};
                            *(d2 *)(p + 16) = d2{x1[0], x1[1]};
                            *(d2 *)(p + 18) = d2{x1[2], x1[3]};
                        }
                    }
                }
            }
        }
        if (ngroups == 0) {
            __syncthreads();
            chol_block_out(pj, linv, Lm, iv, n, j0, w, tid);
        }
        STAMP(); // 8: all tile groups solved and stored (issued)
        if (CHOL_ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads(); // the panel is in memory (and pj / linv are free) before the next one reads it
        STAMP(); // 9: panel end
    }
    if (wave == 0 && lane0 == 0 && bad) status[blockIdx.x] = 1;
}

// ---- round 3: the same factorisation with the panel staged FIRST ------------------------------------------------------------
// Time stamps of one workgroup (tools/chol_probe.hip, s_memtime, uncontended: 276 us per system of order 400, 13 panels) showed
// wave 0 alone on the critical path: 21 % of it in the diagonal block's update (its operands were panel rows fetched from L2
// right after the barrier that publishes them: one exposed memory round trip per panel), 7 % in wave 0 summing the eight partial
// blocks by itself, 36 % in the factor / invert sweep, 18 % in wave 0's own off-diagonal tiles that it starts only after the sweep
// while the other seven waves wait at the barrier.  Here:
//   1. the panel rows L[j0 .. j0+31][0 .. j0) are staged into LDS first (they are needed there anyway);
//   2. the diagonal update D = A_jj - P P^T runs from LDS on all eight waves: 4 output tiles x 2 k-halves, no global operand,
//      two partial blocks instead of eight (the half with k < j0 / 2 starts from the source block, whose loads were issued
//      before the staging);
//   3. wave 0 adds the two partials while it loads its row for the sweep (64 LDS reads instead of 256 + a barrier);
//   4. wave 0 owns no off-diagonal tile while another wave has fewer than two: the sweep is the longest serial piece of a panel.
// One workgroup barrier less per panel.  Results are bitwise those of k_chol_left<true> up to the summation order of the diagonal
// update (two k-halves instead of eight k-ranges).
__device__ __forceinline__ void diag_from_lds(const double *la, const double *lb, int kb, int ke, d4 &acc)
{
    for (int k = kb; k < ke; k += 32) {
        d2 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = *(const d2 *)(la + k + LOFF_V * v);
            b[v] = *(const d2 *)(lb + k + LOFF_V * v);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = MFMA_F64(-a[e >> 1][e & 1], b[e >> 1][e & 1], acc);
    }
}

template <int TW>
__global__ __launch_bounds__(512, 1) void k_chol_left2(int n_, double *Afull, double *invd, long sinv, int *status, const double *Apacked,
                                                       long spk, double diag_add)
{
    __shared__ __attribute__((aligned(16))) double blk[2][32][34]; // the factored diagonal block and its inverse (sweep32)
    double (*pj)[34] = blk[0], (*linv)[34] = blk[1];
    __shared__ __attribute__((aligned(16))) double col[2][64], piv[32];
    __shared__ __attribute__((aligned(16))) double part[2][32][34]; // the two k-halves of the diagonal block's update
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[]; // the panel rows L[j0 .. j0 + 31][0 .. j0)
    double *pan = dyn_lds;
    const int S = pan_stride(n_);
    const long n = n_;
    double *Lm = Afull + (size_t)blockIdx.x * n * n;
    const double *Apk = Apacked ? Apacked + (size_t)blockIdx.x * spk : nullptr;
    double *iv = invd + (size_t)blockIdx.x * sinv;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    int bad = 0;
    d4 acc[TW][2];
    PROF_INIT(0, wave, lane0);

    for (int j0 = 0; j0 < n_; j0 += 32) {
        PROF_PANEL(j0 >> 5);
        STAMP(); // 0: panel start
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int i16 = lane & 15, q = lane >> 4, perm = 4 * (i16 & 3) + (i16 >> 2);
        const int w = (n_ - j0) < 32 ? (n_ - j0) : 32;
        const int below = n_ - j0 - 32;
        const int nt = below > 0 ? (below + 15) / 16 : 0;
        // tile owners: waves 1 2 3 5 6 7 4 take a tile each in turn; wave 0 (the sweep) joins only from the third round on
        int mine;
        {
            const int s7 = wave == 4 ? 6 : (wave < 4 ? wave - 1 : wave - 2); // slot among the seven workers (wave 0: -1)
            const int first = nt < 14 ? nt : 14;                              // the first two rounds go to the workers alone
            const int rest = nt - first;                                      // then all eight waves, wave 0 last
            const int slot8 = wave == 0 ? 7 : s7;
            mine = (wave == 0 ? 0 : (first > s7 ? (first - s7 + 6) / 7 : 0)) + (rest > slot8 ? (rest - slot8 + 7) / 8 : 0);
        }
        const int maxmine = nt <= 14 ? (nt + 6) / 7 : 2 + (nt - 14 + 7) / 8;
        const int ngroups = (maxmine + TW - 1) / TW; // uniform over the workgroup
        // tile index of this wave's m-th tile
        auto tile_of = [&](int m) -> int {
            const int s7 = wave == 4 ? 6 : (wave < 4 ? wave - 1 : wave - 2);
            if (wave != 0 && m < 2) { const int t = s7 + 7 * m; if (t < (nt < 14 ? nt : 14)) return t; }
            const int mm = wave == 0 ? m : m - ((nt < 14 ? nt : 14) > s7 ? (((nt < 14 ? nt : 14) - s7 + 6) / 7) : 0);
            return 14 + (wave == 0 ? 7 : s7) + 8 * mm;
        };

        // ---- the source of the diagonal block (half 0 of each of the four output tiles), requested before the staging ----
        const int du = wave & 1, dct = (wave >> 1) & 1, dh = wave >> 2; // this wave's tile (rows 16 du.., columns 16 dct..) and k-half
        d4 dacc = d4{0.0, 0.0, 0.0, 0.0};
        if (dh == 0) {
            long r = j0 + 16 * du + i16;
            r = r < n ? r : n - 1;
            if (w == 32) dacc = chol_src4(Lm, Apk, n, r, j0 + 16 * dct + 4 * q, diag_add);
            else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = 16 * du + i16, k = 16 * dct + 4 * q + rr;
                    double v = 0.0;
                    if (i < w && k < w) {
                        if (!Apk) v = Lm[(long)(j0 + i) * n + j0 + k];
                        else if (k <= i) v = Apk[(long)(j0 + i) * (j0 + i + 1) / 2 + j0 + k] + (k == i ? diag_add : 0.0);
                    }
                    dacc[rr] = v;
                }
            }
        }
        // ---- 1. panel rows -> LDS ----
        if (j0 > 0) stage_panel(pan, S, Lm, n, j0, 0, j0, tid);
        __syncthreads();
        STAMP(); // 1: staged
        // ---- 2. diagonal update from LDS: tile (du, dct), k-half dh ----
        {
            const int half = ((j0 >> 5) >> 1) << 5; // whole 32-chunks
            const int kb = dh == 0 ? 0 : half, ke = dh == 0 ? half : j0;
            if (!(CHOL_ABL & 8)) diag_from_lds(pan + (16 * dct + perm) * S + LOFF_Q * q, pan + (16 * du + i16) * S + LOFF_Q * q, kb, ke, dacc);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * du + i16, k = 16 * dct + 4 * q + rr;
                part[dh][i][k] = (i < w && k < w) ? dacc[rr] : ((i == k && dh == 0) ? 1.0 : 0.0);
            }
        }
        __syncthreads();
        STAMP(); // 2: the two partial blocks are in LDS
        long ra0 = j0 + perm, ra1 = j0 + 16 + perm;
        ra0 = ra0 < n ? ra0 : n - 1;
        ra1 = ra1 < n ? ra1 : n - 1;
        const double *pa0 = Lm + ra0 * n + KOFF_Q * q, *pa1 = Lm + ra1 * n + KOFF_Q * q;
        // operand rows (pb), output rows and tile count of group g
        auto group_setup = [&](int g, const double *(&pb)[TW], long (&rows)[TW]) __attribute__((always_inline)) -> int {
            int cnt = mine - TW * g;
            cnt = cnt < 0 ? 0 : (cnt > TW ? TW : cnt);
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                const int t = u < cnt ? tile_of(TW * g + u) : 0;
                long r = j0 + 32 + 16L * t + i16;
                rows[u] = u < cnt ? r : n;
                r = r < n ? r : n - 1;
                pb[u] = Lm + r * n + KOFF_Q * q;
            }
            return cnt;
        };
        auto group_src = [&](auto pk, int cnt, const long (&rows)[TW], d4 (&ac)[TW][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < TW; ++u)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (u < cnt) {
                        const long r = rows[u] < n ? rows[u] : n - 1;
                        ac[u][ct] = (CHOL_ABL & 64) ? d4{0.0, 0.0, 0.0, 0.0} : chol_src4t<decltype(pk)::value>(Lm, Apk, n, r, j0 + 16 * ct + 4 * q, diag_add);
                    }
        };
        auto group_solve = [&](int cnt, const long (&rows)[TW], const LinvOps &lo, const d4 (&ac)[TW][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u < cnt) {
                    d4 x0, x1;
                    tile_trsm(lo, ac[u], x0, x1);
                    if (rows[u] < n) {
                        double *p = Lm + rows[u] * n + j0 + 4 * q;
                        *(d2 *)p = d2{x0[0], x0[1]};
                        *(d2 *)(p + 2) = d2{x0[2], x0[3]};
                        *(d2 *)(p + 16) = d2{x1[0], x1[1]};
                        *(d2 *)(p + 18) = d2{x1[2], x1[3]};
                    }
                }
            }
        };
        // ---- 3. wave 0: factor + invert; the others: the k-loops of ALL their tiles (up to MAXG groups; the accumulators wait in
        //         registers), so that the sweep -- the longest serial piece of a panel -- has the whole off-diagonal update beside
        //         it, not only the first tile of every wave (stamps of the previous form, panel 1 of 13 at order 400: sweep 14.3 k
        //         cycles, first k-loops 9.1 k, then 27.5 k for the two remaining tiles of every wave AFTER the barrier).  The sources
        //         of all tiles are requested before the first k-loop: one memory round trip, not one per group. ----
        constexpr int MAXG = (4 + TW - 1) / TW; // four tiles per wave cover orders up to ~500
        d4 accs[MAXG][TW][2];
        const int npre = wave == 0 ? 0 : (ngroups < MAXG ? ngroups : MAXG);
        if (wave == 0) {
            const int li = lane & 31;
            double a[32];
#pragma unroll
            for (int k = 0; k < 32; k += 2) {
                const d2 x = *(const d2 *)&part[0][li][k], y = *(const d2 *)&part[1][li][k];
                a[k] = x[0] + y[0];
                a[k + 1] = x[1] + y[1];
            }
            __builtin_amdgcn_s_setprio(3);
            sweep32(a, lane, col, piv, blk, bad); // factor + invert, results in pj / linv
            __builtin_amdgcn_s_setprio(0);
        } else {
            // one branch on the kind of source around ALL the requests, and no guard per group (a group this wave does not have
            // re-reads tile 0's source -- cached, unused): any branch between the requests makes hipcc wait for each of them
            if (Apk) {
#pragma unroll
                for (int g = 0; g < MAXG; ++g) {
                    const double *pb[TW];
                    long rows[TW];
                    group_setup(g, pb, rows);
                    group_src(std::true_type{}, TW, rows, accs[g]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < MAXG; ++g) {
                    const double *pb[TW];
                    long rows[TW];
                    group_setup(g, pb, rows);
                    group_src(std::false_type{}, TW, rows, accs[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
                if (g < npre) {
                    const double *pb[TW];
                    long rows[TW];
                    const int cnt = group_setup(g, pb, rows);
                    if (!(CHOL_ABL & 4)) rowdot_sel<true, true>(cnt, pa0, pa1, pan + perm * S + LOFF_Q * q, pan + (16 + perm) * S + LOFF_Q * q, pb, 0, j0, 0, accs[g]);
                }
        }
        STAMP(); // 3: (wave 0) block factored and inverted / (others) k-loops done
        __syncthreads(); // inv(L_jj) and L_jj are in LDS
        STAMP(); // 4
        chol_block_out(pj, linv, Lm, iv, n, j0, w, tid);
        if (!(CHOL_ABL & 2)) {
            LinvOps lo;
            linv_ops_load(lo, linv, perm, q);
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
                if (g < npre) {
                    const double *pb[TW];
                    long rows[TW];
                    const int cnt = group_setup(g, pb, rows);
                    group_solve(cnt, rows, lo, accs[g]);
                }
        }
        STAMP(); // 5: the tiles of the first phase solved and stored (issued)
        // ---- 4. what is left: wave 0's own tiles, and groups beyond MAXG (orders above ~500) ----
        for (int g = npre; g < ngroups; ++g) {
            const double *pb[TW];
            long rows[TW];
            const int cnt = group_setup(g, pb, rows);
            if (Apk) group_src(std::true_type{}, cnt, rows, acc);
            else group_src(std::false_type{}, cnt, rows, acc);
            if (!(CHOL_ABL & 4)) rowdot_sel<true, true>(cnt, pa0, pa1, pan + perm * S + LOFF_Q * q, pan + (16 + perm) * S + LOFF_Q * q, pb, 0, j0, 0, acc);
            if (cnt > 0 && !(CHOL_ABL & 2)) {
                LinvOps lo;
                linv_ops_load(lo, linv, perm, q);
                group_solve(cnt, rows, lo, acc);
            }
        }
        STAMP(); // 6: all tile groups solved and stored (issued)
        __syncthreads(); // the panel is in memory (and pj / linv / part / pan are free) before the next one reads it
        STAMP(); // 7: panel end
    }
    if (wave == 0 && lane0 == 0 && bad) status[blockIdx.x] = 1;
}

// U = L^-T (upper triangular, row-major; U[c][i] = (L^-1)[i][c]) from the factor and the inverses of its diagonal
// blocks -- the same panel step as the factorisation with the roles shifted: for the 32 columns i0.. of U,
//   T[c][i] = sum_{k < i0} U[c][k] L[i0 + i][k]   (rows of U x rows of L, k contiguous in both),   U[c][i0..] = -T inv(L_ii)^T
// for the row tiles c < i0.  U[c][k] = 0 for k < c: a tile's k range starts at its own 32-block (whose diagonal
// block is stored with its zeros), and tiles of one wave (128 rows apart) join the k loop one after the other.
// Only the upper triangle (plus the diagonal blocks) of U is written; nothing else of U is ever read.
// Row tiles of a panel are dealt to the waves in SNAKE order: round j hands tile NWV j + slot to the wave with slot = wave (j even) or
// NWV - 1 - wave (j odd).  The k range of a tile shrinks with its row index (k_trinv_left: [tile's block, i0); k_uut: [tile's block, n)), so
// with the plain deal wave 0 always held the longest tile of every round -- 832 against a mean of 674 chunk-columns on the first panel of
// k_uut at order 400 -- and everybody waited for it at the panel's barrier; the snake brings that to 720.  Which wave computes a tile
// does not change its value.  Round j is valid for a prefix of j (tile index grows with j), as the pass structure assumes.
// MEASURED (round 6, profiles/r06/chol_experiments.txt, 1024 systems of order 400): k_trinv_left 0.772 -> 0.751 ms, k_uut 1.048 -> 1.024 ms.
// The other round-6 experiment -- TWO independent 4-wave workgroups per CU, each on its own system, so that the hardware interleaves the
// phases of two systems (panel rows per wave from L2: two 100 KB panels do not fit LDS) -- lost: 1.035 / 1.431 ms; the LDS copy of the
// panel rows is worth more than the interleaving (the 8-wave kernels WITHOUT it: 1.122 / 1.395).  Not kept in the product.
#ifndef CHOL_SNAKE
#define CHOL_SNAKE 1
#endif
template <int NWV> __device__ __forceinline__ int snake_slot(int wave, int j) { return (CHOL_SNAKE && (j & 1)) ? NWV - 1 - wave : wave; }
template <int NWV> __device__ __forceinline__ int snake_count(int wave, int nt)
{
    const int full = nt / NWV; // rounds in which every wave has a tile
    return full + (snake_slot<NWV>(wave, full) < nt - NWV * full ? 1 : 0);
}

template <bool use_lds, int TW, int NWV = 8>
__global__ __launch_bounds__(NWV * 64, 1) void k_trinv_left(int n_, const double *__restrict__ Lfull, const double *__restrict__ invd,
                                                       long sinv, double *Ufull)
{
    __shared__ __attribute__((aligned(16))) double linv[32][34];
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[]; // use_lds: the panel rows L[i0 .. i0 + 31][0 .. i0)
    double *pan = dyn_lds;
    const int S = pan_stride(n_);
    const long n = n_;
    const double *Lm = Lfull + (size_t)blockIdx.x * n * n;
    double *Um = Ufull + (size_t)blockIdx.x * n * n;
    const double *iv = invd + (size_t)blockIdx.x * sinv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int perm = 4 * (i16 & 3) + (i16 >> 2);
    d4 acc[TW][2];
    for (int i0 = 0; i0 < n_; i0 += 32) {
        const int w = (n_ - i0) < 32 ? (n_ - i0) : 32;
        for (int e = tid; e < 1024; e += NWV * 64) {
            const int i = e >> 5, k = e & 31;
            const double v = iv[(size_t)(i0 >> 5) * 1024 + e];
            linv[i][k] = v;
            if (i < w && k < w) Um[(long)(i0 + k) * n + i0 + i] = v; // diagonal block of U = inv(L_ii)^T
        }
        if (use_lds && i0 > 0) stage_panel<2 * NWV>(pan, S, Lm, n, i0, 0, i0, tid);
        __syncthreads();
        const int nt = i0 >> 4; // full row tiles above the diagonal block
        const int mine = snake_count<NWV>(wave, nt);
        long ra0 = i0 + perm, ra1 = i0 + 16 + perm;
        ra0 = ra0 < n ? ra0 : n - 1;
        ra1 = ra1 < n ? ra1 : n - 1;
        const double *pa0 = Lm + ra0 * n + KOFF_Q * q, *pa1 = Lm + ra1 * n + KOFF_Q * q;
        for (int g = 0; TW * g < mine; ++g) {
            int cnt = mine - TW * g;
            cnt = cnt > TW ? TW : cnt;
            const double *pb[TW];
            long rows[TW];
            int ks[TW + 1];
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                long r0 = 16L * (snake_slot<NWV>(wave, TW * g + u) + NWV * (TW * g + u));
                r0 = u < cnt ? r0 : 16L * wave; // unused slots alias a valid tile
                rows[u] = r0 + i16;
                pb[u] = Um + rows[u] * n + KOFF_Q * q;
                ks[u] = (int)(r0 >> 5) << 5;
                acc[u][0] = acc[u][1] = d4{0.0, 0.0, 0.0, 0.0};
            }
            ks[TW] = i0;
#pragma unroll
            for (int s = 0; s < TW; ++s)
                if (s < cnt)
                    rowdot_sel<use_lds, true>(s + 1, pa0, pa1, pan + perm * S + LOFF_Q * q, pan + (16 + perm) * S + LOFF_Q * q, pb, ks[s],
                                              (s + 1 < cnt) ? ks[s + 1] : i0, 0, acc);
            LinvOps lo;
            linv_ops_load(lo, linv, perm, q);
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u < cnt) {
                    d4 x0, x1;
                    tile_trsm(lo, acc[u], x0, x1);
                    double *p = Um + rows[u] * n + i0 + 4 * q;
                    if (i0 + 4 * q < n) *(d2 *)p = d2{x0[0], x0[1]};
                    if (i0 + 4 * q + 2 < n) *(d2 *)(p + 2) = d2{x0[2], x0[3]};
                    if (i0 + 16 + 4 * q < n) *(d2 *)(p + 16) = d2{x1[0], x1[1]};
                    if (i0 + 18 + 4 * q < n) *(d2 *)(p + 18) = d2{x1[2], x1[3]};
                }
            }
        }
        __syncthreads(); // the 32 columns are in memory (and linv is free) before the next panel
    }
}

// inv = U U^T = A^-1:  inv[i][j] = sum_{k >= max(i, j)} U[i][k] U[j][k]  -- rows x rows again.  Panels of 32 columns
// j0.., row tiles i >= j0, k from the tile's own 32-block to n (a last partial chunk is masked).  Each tile is
// written to both triangles.  No synchronisation at all: U is only read.
template <int CNT, int TWT>
__device__ __forceinline__ void rowdot_tail(const double *pa0, const double *pa1, const double *(&pb)[TWT], int kt, int n, int q,
                                            d4 (&acc)[TWT][2])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int eo = KOFF_V * (e >> 1) + (e & 1); // element e of the lane's eight: column kt + eo + KOFF_Q q
        const bool ok = kt + eo + KOFF_Q * q < n;
        const double x0 = ok ? pa0[kt + eo] : 0.0, x1 = ok ? pa1[kt + eo] : 0.0;
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const double y = ok ? pb[u][kt + eo] : 0.0;
            acc[u][0] = MFMA_F64(x0, y, acc[u][0]);
            acc[u][1] = MFMA_F64(x1, y, acc[u][1]);
        }
    }
}
// packed != NULL: instead of the full matrix, E = inv + w w^T goes out as PACKED lower rows (stride sp) -- what the T-matrix
// E-step accumulates (A_c += N_uc E_u): no full inverse in memory, no matrix-vector pass, no pack pass.
template <bool use_lds, int TW, int NWV = 8>
__global__ __launch_bounds__(NWV * 64, 1) void k_uut(int n_, const double *__restrict__ Ufull, double *__restrict__ inv,
                                                const double *__restrict__ wv, double *__restrict__ packed, long sp)
{
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[]; // use_lds: the panel rows U[j0 .. j0 + 31][j0 .. nfl)
    double *pan = dyn_lds;
    const int S = pan_stride(n_);
    const long n = n_;
    const double *Um = Ufull + (size_t)blockIdx.x * n * n;
    double *Om = inv ? inv + (size_t)blockIdx.x * n * n : nullptr;
    double *Pk = packed ? packed + (size_t)blockIdx.x * sp : nullptr;
    const double *wm = wv ? wv + (size_t)blockIdx.x * n : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int perm = 4 * (i16 & 3) + (i16 >> 2);
    const int nfl = n_ & ~31;
    d4 acc[TW][2];
    PROF_INIT(2, wave, lane);
    for (int j0 = 0; j0 < n_; j0 += 32) {
        PROF_PANEL(j0 >> 5);
        STAMP(); // 0
        const int nt = (n_ - j0 + 15) >> 4; // row tiles from the diagonal block down
        const int mine = snake_count<NWV>(wave, nt);
        long ra0 = j0 + perm, ra1 = j0 + 16 + perm;
        ra0 = ra0 < n ? ra0 : n - 1;
        ra1 = ra1 < n ? ra1 : n - 1;
        const double *pa0 = Um + ra0 * n + KOFF_Q * q, *pa1 = Um + ra1 * n + KOFF_Q * q;
        if (use_lds) {
            if (j0 > 0) __syncthreads(); // every wave is done with the previous panel's rows
            if (nfl > j0) stage_panel<2 * NWV>(pan, S, Um, n, j0, j0, nfl - j0, tid);
            __syncthreads();
        }
        STAMP(); // 1: staged
        for (int g = 0; TW * g < mine; ++g) {
            int cnt = mine - TW * g;
            cnt = cnt > TW ? TW : cnt;
            const double *pb[TW];
            long rows[TW];
            int ks[TW + 1];
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                long r0 = j0 + 16L * (snake_slot<NWV>(wave, TW * g + u) + NWV * (TW * g + u));
                r0 = u < cnt ? r0 : j0 + 16L * wave;
                rows[u] = r0 + i16;
                const long rc = rows[u] < n ? rows[u] : n - 1;
                pb[u] = Um + rc * n + KOFF_Q * q;
                const int k0 = (int)(r0 >> 5) << 5;
                ks[u] = k0 < nfl ? k0 : nfl;
                acc[u][0] = acc[u][1] = d4{0.0, 0.0, 0.0, 0.0};
            }
            ks[TW] = nfl;
#pragma unroll
            for (int s = 0; s < TW; ++s)
                if (s < cnt)
                    rowdot_sel<use_lds, false>(s + 1, pa0, pa1, pan + perm * S + LOFF_Q * q, pan + (16 + perm) * S + LOFF_Q * q, pb, ks[s],
                                               (s + 1 < cnt) ? ks[s + 1] : nfl, j0, acc);
            if (nfl < n_) {
                switch (cnt) {
                case 1: rowdot_tail<1>(pa0, pa1, pb, nfl, n_, q, acc); break;
                case 2: rowdot_tail<(TW >= 2 ? 2 : TW)>(pa0, pa1, pb, nfl, n_, q, acc); break;
                case 3: rowdot_tail<(TW >= 3 ? 3 : TW)>(pa0, pa1, pb, nfl, n_, q, acc); break;
                default: break;
                }
            }
            STAMP(); // k-loops of a group done
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u < cnt && rows[u] < n && Pk) {
                    const long row = rows[u];
                    const double wr = wm ? wm[row] : 0.0;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const long c0 = j0 + 16 * ct + 4 * q;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r <= row) Pk[row * (row + 1) / 2 + c0 + r] = acc[u][ct][r] + (wm ? wr * wm[c0 + r] : 0.0);
                    }
                } else if (u < cnt && rows[u] < n) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const long c0 = j0 + 16 * ct + 4 * q;
                        double *p = Om + rows[u] * n + c0;
                        if (c0 < n) *(d2 *)p = d2{acc[u][ct][0], acc[u][ct][1]};
                        if (c0 + 2 < n) *(d2 *)(p + 2) = d2{acc[u][ct][2], acc[u][ct][3]};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r < n) Om[(c0 + r) * n + rows[u]] = acc[u][ct][r];
                    }
                }
            }
            STAMP(); // stores of a group issued
        }
    }
}

// X = A^-1 B for NR <= 64 right-hand sides per system through the Cholesky factor (k_chol_left) and the inverses of its 32 x 32
// diagonal blocks: blocked forward (L Y = B) and backward (L^T X = Y) substitution on the matrix cores, one workgroup per system.
// TVAcc::updateTestimate (AccumulateTVStat.cpp:981-1000) is T_c = A_c^-1 Cmx_c with 60 columns: the reference inverts A_c
// explicitly; substituting needs 2 n^2 NR flops instead of the 4 n^3 / 3 of triangular inverse + U U^T and leaves out two of the
// three kernels per batch.  8 waves = 2 row halves x 4 column tiles of a 32-row block; Y / X live in the output array itself
// (element (k, d) at X[k ldx + d]) and are re-read from there (L2) as the B operand of later block rows -- visible to the other
// waves of the workgroup after the barrier, like the panels of k_chol_left.
__global__ __launch_bounds__(512, 1) void k_chol_solve_multi(int n_, int NR, const double *__restrict__ Lfull, const double *__restrict__ invd,
                                                             long sinv, const double *__restrict__ Bsrc, long ldb, long sB,
                                                             double *__restrict__ Xdst, long ldx, long sX)
{
    __shared__ __attribute__((aligned(16))) double rs[32][66];  // the block row's right-hand side after the update
    __shared__ __attribute__((aligned(16))) double dinv[32][34]; // inv(L_ii)
    const long n = n_;
    const double *Lm = Lfull + (size_t)blockIdx.x * n * n;
    const double *iv = invd + (size_t)blockIdx.x * sinv;
    const double *Bm = Bsrc + (size_t)blockIdx.x * sB;
    double *Xm = Xdst + (size_t)blockIdx.x * sX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int rt = wave >> 2, ct = wave & 3;
    const int col = 16 * ct + i16;          // right-hand side of this lane (B operand / result column)
    const bool colok = col < NR;
    const int colc = colok ? col : NR - 1;
    const int nblk = (n_ + 31) >> 5;
    // ---- forward: L Y = B, block rows top down ----
    for (int ib = 0; ib < nblk; ++ib) {
        const int r0 = ib << 5;
        for (int e = tid; e < 1024; e += 512) dinv[e >> 5][e & 31] = iv[(size_t)ib * 1024 + e];
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = r0 + 16 * rt + q + 4 * r;
            acc[r] = (row < n && colok) ? Bm[row * ldb + col] : 0.0;
        }
        long ar = r0 + 16 * rt + i16;        // row of L this lane feeds as the A operand
        ar = ar < n ? ar : n - 1;
        const double *pa = Lm + ar * n + 8 * q;
        for (int k = 0; k < r0; k += 32) {
            d2 a[4];
            double b[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) a[v] = *(const d2 *)(pa + k + 2 * v);
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = Xm[(long)(k + 8 * q + e) * ldx + colc];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = MFMA_F64(-a[e >> 1][e & 1], b[e], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[16 * rt + q + 4 * r][col] = acc[r];
        __syncthreads();
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sk = 0; sk < 8; ++sk) y = MFMA_F64(dinv[16 * rt + i16][4 * sk + q], rs[4 * sk + q][col], y);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = r0 + 16 * rt + q + 4 * r;
            if (row < n && colok) Xm[row * ldx + col] = y[r];
        }
        __syncthreads(); // Y of this block row is in memory, rs / dinv are free
    }
    // ---- backward: L^T X = Y, block rows bottom up ----
    for (int ib = nblk - 1; ib >= 0; --ib) {
        const int c0 = ib << 5;
        for (int e = tid; e < 1024; e += 512) dinv[e >> 5][e & 31] = iv[(size_t)ib * 1024 + e];
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = c0 + 16 * rt + q + 4 * r;
            acc[r] = (row < n && colok) ? Xm[row * ldx + col] : 0.0;
        }
        long ac = c0 + 16 * rt + i16;        // column of L = row of L^T this lane feeds as the A operand
        ac = ac < n ? ac : n - 1;
        for (int k = c0 + 32; k < n_; k += 32) {
            double a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long kk = k + 8 * q + e;
                const bool ok = kk < n;
                const long kc = ok ? kk : n - 1;
                const double av = Lm[kc * n + ac], bv = Xm[kc * ldx + colc];
                a[e] = ok ? av : 0.0;
                b[e] = ok ? bv : 0.0;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = MFMA_F64(-a[e], b[e], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[16 * rt + q + 4 * r][col] = acc[r];
        __syncthreads();
        d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sk = 0; sk < 8; ++sk) x = MFMA_F64(dinv[4 * sk + q][16 * rt + i16], rs[4 * sk + q][col], x); // inv(L_ii)^T
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = c0 + 16 * rt + q + 4 * r;
            if (row < n && colok) Xm[row * ldx + col] = x[r];
        }
        __syncthreads();
    }
}

int tvk_chol_solve_multi_batched(hipStream_t st, int n, int nb, int nrhs, const double *Lf, const double *invd, const double *B, long ldb,
                                 long sB, double *X, long ldx, long sX)
{
    if (nb <= 0 || n <= 0 || nrhs <= 0) return 0;
    if (nrhs > 64 || (n & 1)) return -1; // the caller keeps the explicit inverse
    k_chol_solve_multi<<<nb, 512, 0, st>>>(n, nrhs, Lf, invd, (long)((n + 31) / 32) * 1024, B, ldb, sB, X, ldx, sX);
    return (int)hipGetLastError();
}

// Dynamic LDS of the three kernels and whether the panel rows fit next to the static arrays (160 KB per workgroup on gfx950)
#ifndef CHOL_TW_NOLDS
#define CHOL_TW_NOLDS 2 // row tiles per wave and pass of the kernels that fetch the panel rows per wave (orders whose panel does not fit LDS): order 600, 512 systems, trinv + U U^T 4.15 / 4.09 / 4.81 ms for 3 / 2 / 1
#endif
namespace {
struct CholLds { int use; size_t chol, trinv, uut; };
CholLds chol_lds(int n)
{
    const size_t pan = (size_t)32 * ((n & 2) ? n : n + 2) * sizeof(double), slab = (size_t)8 * 32 * 33 * sizeof(double);
    CholLds r;
    r.use = g_chol_lds && pan + 36 * 1024 <= 160 * 1024 ? 1 : 0; // static LDS of k_chol_left2: pj, linv, col, part = 35 KB
    r.chol = r.use && pan > slab ? pan : slab;
    r.trinv = r.use ? pan : 16;
    r.uut = r.use ? pan : 16;
    return r;
}
#ifndef CHOL_TW_FLOW
#define CHOL_TW_FLOW 1 // tiles per wave and k-loop of k_chol_left2 (tools/chol_probe.sh build -DCHOL_TW_FLOW=2 for A/B runs)
#endif
int launch_chol(hipStream_t st, int n, int nb, double *Afull, double *invd, int *status, const double *Apacked, long spk, double diag_add)
{
    prof_host_init();
    const CholLds l = chol_lds(n);
    const long sinv = (long)((n + 31) / 32) * 1024;
    if (l.use && gmmiv_kopts_cur().chol_flow) { // round 3: panel staged first, diagonal update from LDS (k_chol_left2)
        const size_t pan = (size_t)32 * ((n & 2) ? n : n + 2) * sizeof(double);
        int rc2 = (int)gmmiv_lds_attr<k_chol_left2<CHOL_TW_FLOW>>(pan);
        if (rc2) return rc2;
        k_chol_left2<CHOL_TW_FLOW><<<nb, 512, pan, st>>>(n, Afull, invd, sinv, status, Apacked, spk, diag_add);
        return (int)hipGetLastError();
    }
    int rc = l.use ? (int)gmmiv_lds_attr<k_chol_left<true>>(l.chol) : (int)gmmiv_lds_attr<k_chol_left<false>>(l.chol);
    if (rc) return rc;
    if (l.use) k_chol_left<true><<<nb, 512, l.chol, st>>>(n, Afull, invd, sinv, status, Apacked, spk, diag_add);
    else k_chol_left<false><<<nb, 512, l.chol, st>>>(n, Afull, invd, sinv, status, Apacked, spk, diag_add);
    return (int)hipGetLastError();
}
int launch_trinv(hipStream_t st, int n, int nb, const double *Lf, const double *invd, double *U)
{
    const CholLds l = chol_lds(n);
    const long sinv = (long)((n + 31) / 32) * 1024;
    if (l.use && gmmiv_kopts_cur().chol_waves == 16) { // 16 waves of 128 VGPRs, one row tile per wave and pass: twice the waves to cover a stalled one
        int rc16 = (int)gmmiv_lds_attr<k_trinv_left<true, 1, 16>>(l.trinv);
        if (rc16) return rc16;
        k_trinv_left<true, 1, 16><<<nb, 1024, l.trinv, st>>>(n, Lf, invd, sinv, U);
        return (int)hipGetLastError();
    }
    int rc = l.use ? (int)gmmiv_lds_attr<k_trinv_left<true, 1>>(l.trinv) : (int)gmmiv_lds_attr<k_trinv_left<false, CHOL_TW_NOLDS>>(l.trinv);
    if (rc) return rc;
    if (l.use) k_trinv_left<true, 1><<<nb, 512, l.trinv, st>>>(n, Lf, invd, sinv, U);
    else k_trinv_left<false, CHOL_TW_NOLDS><<<nb, 512, l.trinv, st>>>(n, Lf, invd, sinv, U);
    return (int)hipGetLastError();
}
int launch_uut(hipStream_t st, int n, int nb, const double *U, double *inv, const double *w, double *packed, long sp)
{
    const CholLds l = chol_lds(n);
    if (l.use && gmmiv_kopts_cur().chol_waves == 16) {
        int rc16 = (int)gmmiv_lds_attr<k_uut<true, 1, 16>>(l.uut);
        if (rc16) return rc16;
        k_uut<true, 1, 16><<<nb, 1024, l.uut, st>>>(n, U, inv, w, packed, sp);
        return (int)hipGetLastError();
    }
    int rc = l.use ? (int)gmmiv_lds_attr<k_uut<true, 2>>(l.uut) : (int)gmmiv_lds_attr<k_uut<false, CHOL_TW_NOLDS>>(l.uut);
    if (rc) return rc;
    if (l.use) k_uut<true, 2><<<nb, 512, l.uut, st>>>(n, U, inv, w, packed, sp);
    else k_uut<false, CHOL_TW_NOLDS><<<nb, 512, l.uut, st>>>(n, U, inv, w, packed, sp);
    return (int)hipGetLastError();
}
} // namespace

int tvk_chol_left_batched(hipStream_t st, int n, int nb, double *Afull, double *invd, int *status, const double *Apacked, long spk,
                          double diag_add)
{
    if (nb <= 0 || n <= 0) return 0;
    return launch_chol(st, n, nb, Afull, invd, status, Apacked, spk, diag_add);
}
// the other two steps on their own (tools/chol_probe.hip times them one by one)
int tvk_trinv_left_batched(hipStream_t st, int n, int nb, const double *Lf, const double *invd, double *U)
{
    if (nb <= 0 || n <= 0) return 0;
    return launch_trinv(st, n, nb, Lf, invd, U);
}
int tvk_uut_packed_batched(hipStream_t st, int n, int nb, const double *U, const double *w, double *P, long sp)
{
    if (nb <= 0 || n <= 0) return 0;
    return launch_uut(st, n, nb, U, nullptr, w, P, sp);
}

// inv[b] = A[b]^-1 through the three one-workgroup-per-matrix kernels; U: scratch nb*n*n (only its upper triangle is used).
// Apacked != NULL: A comes as packed lower rows (stride spk) + diag_add I, Afull only receives the factor.
int tvk_spd_inverse_left_batched(hipStream_t st, int n, int nb, double *Afull, double *inv, double *U, double *invd, int *status,
                                 const double *Apacked, long spk, double diag_add)
{
    if (nb <= 0 || n <= 0) return 0;
    int rc;
    if ((rc = launch_chol(st, n, nb, Afull, invd, status, Apacked, spk, diag_add))) return rc;
    if ((rc = launch_trinv(st, n, nb, Afull, invd, U))) return rc;
    return launch_uut(st, n, nb, U, inv, nullptr, nullptr, 0);
}

// T-matrix E-step form: A[b] arrives as packed lower rows + diag_add I in P[b] (stride sp); on return W[b] = A^-1 aux[b] and
// P[b] holds E[b] = A^-1 + w w^T, packed.  Lf / U: scratch nb*n*n each (factor, triangular inverse).
int tvk_inverse_e_packed_batched(hipStream_t st, int n, int nb, double *Lf, double *U, double *invd, int *status, double *P, long sp,
                                 double diag_add, const double *aux, double *W)
{
    if (nb <= 0 || n <= 0) return 0;
    int rc;
    if ((rc = launch_chol(st, n, nb, Lf, invd, status, P, sp, diag_add))) return rc;
    if ((rc = tvk_chol_solve_batched(st, n, nb, Lf, invd, aux, W))) return rc;
    if ((rc = launch_trinv(st, n, nb, Lf, invd, U))) return rc;
    return launch_uut(st, n, nb, U, nullptr, W, P, sp);
}
