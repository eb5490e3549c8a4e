#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 PMC summaries of tools/profile_r06.sh (gpurun_out/prof_r06/*pmc*.txt, written by
tools/rocpd_summary.py --pmc: one row per kernel and counter with CALLS and the AVERAGE counter value per launch).
HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (both counters count KB; FETCH_SIZE x 2 is the gfx950 correction of
MI355X_MICROARCH.md's HBM section, calibrated with tools/calib_fetch.py in round 1).  The file records the sha256 of the libgmmiv.so
the passes ran on (written on the GPU box) and the git revision of the tree that was sent: bench.py quotes these figures only when
the library it loads has the same sha256.   usage: python tools/make_traffic.py [gpurun_out/prof_r06] [profiles/r06]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r06")
ref = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06"


def rows(name):
    """{kernel name (as printed): (calls, avg counter value, avg ns)}"""
    out = {}
    for line in open(os.path.join(src, name)):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+(\d+)\s", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    return out


def total(tab, pred=lambda k: True):
    return sum(c * v for k, (c, v, _) in tab.items() if pred(k)) * 1024.0


def one(tab, prefix):
    ks = [k for k in tab if k.startswith(prefix)]
    assert len(ks) == 1, (prefix, ks)
    return tab[ks[0]]


sha = open(os.path.join(src, "libgmmiv_sha256.txt")).read().strip()
rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "lia_ral_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
T = {"_comment": "HBM bytes from the rocprofv3 PMC passes of round 6 (tools/profile_r06.sh -> %s/*pmc*.txt -> tools/make_traffic.py). FETCH_SIZE and "
                 "WRITE_SIZE count KB; FETCH_SIZE x 2 is the gfx950 correction.  bench.py quotes a figure only when the libgmmiv.so it loads has the "
                 "sha256 recorded here." % ref,
     "libgmmiv_sha256": sha, "git_rev": rev + ("+uncommitted kernel changes" if dirty else ""),
     "source": "%s/bench_em_pmc_fetch_size.txt, %s/bench_em_pmc_write_size.txt (tools/profile_r06.sh)" % (ref, ref)}

# ---- EM headline: 10 M frames, (warm-up + steps) iterations of the same launches --------------------------------------------
f, w = rows("bench_em_pmc_fetch_size.txt"), rows("bench_em_pmc_write_size.txt")
C, D, frames = 2048, 60, 10_000_000
k1f, k1w = one(f, "k_llk_mfma<15, float, 8, 1>"), one(w, "k_llk_mfma<15, float, 8, 1>")
k2f, k2w = one(f, "k_stats_z<15, true, float"), one(w, "k_stats_z<15, true, float")
iters = 3
launches = k1f[0] // iters
fpl = frames / launches
T["frames_per_launch"] = fpl
T["algorithmic_bytes_per_launch"] = fpl * 248.0
T["_algorithmic"] = "SURVEY.md 8(d): the feature stream read once + the per-frame log-likelihood: 248 B per frame, independent of the number of Gaussians."
T["_design"] = ("k_llk_mfma<WZ> writes every scaled likelihood once (8 B per frame-Gaussian pair) and k_stats_z reads it back: ~69 x the algorithmic bytes per "
                "kernel, by design (measured faster than recomputing the logits, DESIGN.md section 3); both kernels stay MFMA-bound.")
T["design_bytes_per_pair"] = 16
T["k_llk_mfma_hbm_bytes_per_launch"] = (k1f[1] * 2 + k1w[1]) * 1024.0
T["k_llk_mfma_written_bytes_per_launch"] = k1w[1] * 1024.0
T["k_llk_mfma_ms_under_pmc"] = k1f[2] * 1e-6
T["k_llk_mfma_ratio_to_algorithmic"] = T["k_llk_mfma_hbm_bytes_per_launch"] / T["algorithmic_bytes_per_launch"]
T["k_stats_z_hbm_bytes_per_launch"] = (k2f[1] * 2 + k2w[1]) * 1024.0
T["k_stats_z_ratio_to_algorithmic"] = T["k_stats_z_hbm_bytes_per_launch"] / T["algorithmic_bytes_per_launch"]
T["em_pass_ratio_to_algorithmic"] = T["k_llk_mfma_ratio_to_algorithmic"] + T["k_stats_z_ratio_to_algorithmic"]

# ---- IvExtractor: PASSES passes over U utterances (tools/pmc_blocks.py iv U PASSES): every kernel of the library counts -----
U, PASSES = 2560, 2
f, w = rows("iv_pmc_fetch_size.txt"), rows("iv_pmc_write_size.txt")
lib = lambda k: k.startswith(("k_", "tvk_")) and not k.startswith(("k_tett_packed", "k_gmm_"))      # setup (TETt once, model pack) and torch's kernels excluded
by = {}
for k in set(f) | set(w):
    if lib(k):
        by[k] = (f.get(k, (0, 0, 0))[0] * f.get(k, (0, 0, 0))[1] * 2 + w.get(k, (0, 0, 0))[0] * w.get(k, (0, 0, 0))[1]) * 1024.0 / (U * PASSES)
tot = sum(by.values())
T["iv_extractor"] = {"hbm_bytes_per_utterance": tot, "utterances": U, "passes": PASSES,
                     "algorithmic_bytes_per_utterance": 3000 * 60 * 4 + 400 * 8.0,
                     "ratio_to_algorithmic": tot / (3000 * 60 * 4 + 400 * 8.0),
                     "largest_kernels_bytes_per_utterance": dict(sorted(by.items(), key=lambda kv: -kv[1])[:8]),
                     "source": "%s/iv_pmc_fetch_size.txt, %s/iv_pmc_write_size.txt (python tools/pmc_blocks.py iv %d %d)" % (ref, ref, U, PASSES)}

# ---- T-matrix EM: bench.py --workload tv --tv-utterances 2048, 3 iterations; the statistics pass (K1 / K3, once) excluded -----
U, ITER = 2048, 3
f, w = rows("bench_tv_pmc_fetch_size.txt"), rows("bench_tv_pmc_write_size.txt")
it = lambda k: k.startswith(("k_", "tvk_")) and not k.startswith(("k_llk_mfma", "k_stats_z", "k_flag_frames", "k_gmm_", "k_llk_finalize", "k_count_dead"))
by = {}
for k in set(f) | set(w):
    if it(k):
        by[k] = (f.get(k, (0, 0, 0))[0] * f.get(k, (0, 0, 0))[1] * 2 + w.get(k, (0, 0, 0))[0] * w.get(k, (0, 0, 0))[1]) * 1024.0 / (U * ITER)
tot = sum(by.values())
est = lambda k: not k.startswith(("k_tett_packed", "k_chol_solve_multi", "k_subtract_m", "tvk_subtract", "k_md_", "k_lower"))
T["tv_em"] = {"iteration_hbm_bytes_per_utterance": tot, "estep_hbm_bytes_per_utterance": sum(v for k, v in by.items() if est(k)),
              "utterances": U, "iterations": ITER, "algorithmic_bytes_per_utterance": (C * D + C) * 8.0,
              "largest_kernels_bytes_per_utterance": dict(sorted(by.items(), key=lambda kv: -kv[1])[:10]),
              "source": "%s/bench_tv_pmc_fetch_size.txt, %s/bench_tv_pmc_write_size.txt (bench.py --workload tv --tv-utterances %d, %d iterations)" % (ref, ref, U, ITER)}
# the Cholesky family per 1024 systems (verdict item 4): launches of 1024 systems (2 batches per iteration at 2048 utterances)
for name, key in (("k_chol_left2", "k_chol_left2"), ("k_trinv_left", "k_trinv_left<"), ("k_uut", "k_uut<")):
    ks = [k for k in f if k.startswith(key)]
    if ks:
        T["tv_em"][name + "_fetched_bytes_per_launch"] = f[ks[0]][1] * 2 * 1024.0
        T["tv_em"][name + "_ms_under_pmc"] = f[ks[0]][2] * 1e-6

# ---- scoring: PASSES Mahalanobis calls at M x M --------------------------------------------------------------------------------
M, PASSES = 100_000, 2
f, w = rows("score_pmc_fetch_size.txt"), rows("score_pmc_write_size.txt")
lib = lambda k: k.startswith(("k_", "tvk_"))
tot = (total(f, lib) * 2 + total(w, lib)) / PASSES
T["scoring"] = {"mahalanobis_hbm_bytes_per_call": tot, "models": M, "segments": M, "algorithmic_bytes_per_call": M * M * 8.0 + 2 * M * 400 * 8.0,
                "ratio_to_algorithmic": tot / (M * M * 8.0 + 2 * M * 400 * 8.0),
                "source": "%s/score_pmc_fetch_size.txt, %s/score_pmc_write_size.txt (python tools/pmc_blocks.py score %d %d)" % (ref, ref, M, PASSES)}
json.dump(T, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in T.items() if not k.startswith("_")}, indent=1)[:3000])
