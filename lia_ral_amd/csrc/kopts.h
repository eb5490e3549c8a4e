// kopts.h -- the options of a context that its kernel launchers read (internal).
#pragma once

// Options that reach the kernel launchers (stats_z.hip, tv_kernels.hip, chol_fused.hip), which see a stream, not a context.
// They LIVE in the context (gmmiv_ctx::ko); a call binds its context's set to the calling thread on entry (GBIND, together
// with hipSetDevice) and the launchers read the bound set -- two contexts driven from one thread keep their own settings, one
// context driven from two threads shows both the same.
struct gmmiv_kopts {
    int z_waves = 8;      // workgroup shape of k_stats_z: 8 (one workgroup per CU), 16, or 4 (two per CU)
    int z_tv4 = 1;        // 0 = two Gaussian tiles per wave in the N / F mode of k_stats_z too
    int z_depth_em = 2, z_depth_tv = 4; // stream register sets of k_stats_z per mode (measured: EM 30.5 ms (2) / 31.1 (4) per 4 M frames, N / F 14.2 (four tiles, 2) / 13.7 (two tiles, 4) per 3 M)
    int gemm_remap = 1;   // XCD-aware tile order in k_dgemm: 1 = an XCD walks the M tiles of its N-tile columns fastest; 2 = in 8 x 8 tile blocks (A/B, round 5: same FETCH_SIZE, same time); 0 = hardware order
    int gemm_clamp = 1;   // 0 = cut tiles always on the per-element checked instantiation
    int gemm_narrow = 1;  // 0 = 128 x 128 tiles on the strips cut by M / N too
    int short_calls = 1;  // log-likelihood kernels: calls of at most 32768 frames on 4-wave workgroups (half the latency); 0 = the 8-wave shape of long calls
    int gemm_nt80 = 1;    // split-K NT products with N a multiple of 80 (not of 128) on full row tiles: 128 x 80 tiles, no strip; 0: 128 x 128 + strip
    int chol_lds = 1;     // chol_fused.hip stages the panel rows once per workgroup in LDS; 0: every wave fetches them itself
    int chol_gemm = 0;    // 1 = the GEMM-built right-looking factorisation for every order
    // k_trinv_left / k_uut: 8 waves of 256 VGPRs, or (A/B) 16 waves of 128, one row tile per wave and pass -- twice the waves per SIMD
    // to cover a stalled one, but 30 / 66 spilled VGPRs and twice the LDS operand reads: 0.80 -> 0.96 and 1.13 -> 1.67 ms per 1024 systems
    int chol_waves = 8;
    int chol_flow = 1;    // 1: k_chol_left2 (panel staged first, diagonal update from LDS, wave 0 last in line for tiles); 0: k_chol_left (round 2)
};
const gmmiv_kopts &gmmiv_kopts_cur();          // the set bound to this thread (the defaults before any call)
void gmmiv_kopts_bind(const gmmiv_kopts *ko);  // nullptr: back to the defaults

