"""Pins the CPU oracle against the reference's own known-answer fixtures.

KAT-1  LIA_SpkDet/ComputeTest/test/test1.validate.res  (LLK, top-10, COMPLETE, inclusive end)
KAT-2  LIA_SpkDet/TrainTarget/test/test1.validate.gmm  (full-posterior E-step + MAPOccDep means)
KAT-4  LIA_SpkDet/NormFeat/test/test1.validate.prm     (FrameAccGD mean / biased std)
KAT-5  LIA_Utils/GmmTokenizer/test/test1.sym.ref, mce_matrix.mat.ref  (INTEGERS: top-1 stream, top-20 confusion counts)
KAT-6  LIA_SpkDet/EnergyDetector/test/test1.validate.enr.lbl  (ASSUMED, not a pin: full EM with variances from a fixed init)
Fixtures are the repaired arrays written by tests/golden/make_fixtures.py.
"""
import os

import numpy as np

from oracle import oracle as orc


def _llr(k, complete, ctop):
    world = orc.Gmm(k["w"], k["mean_world"], k["covinv"])
    client = orc.Gmm(k["w_client"], k["mean_client"], k["covinv_client"])
    x = k["x"].astype(np.float64)
    out = []
    for b, n in zip(k["seg_begin"], k["seg_len"]):
        xs = x[b:b + n]
        if ctop is None:
            lw = orc.llk(world, xs); lc = orc.llk(client, xs)
        else:
            d = orc.llk_determine_top(world, xs, ctop, complete)
            lw = d["llk"]
            lc = orc.llk_use_top(client, xs, d["idx"], d["nontop_lk"], complete)
        out.append(lc.mean() - lw.mean())
    return np.array(out)


def test_kat1_computetest_llr(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))
    got = _llr(k, True, int(k["top_c"]))
    assert np.allclose(got, k["expected_llr"], atol=float(k["abs_tol"]), rtol=0), got
    # the fixture discriminates the scoring modes
    partial = _llr(k, False, int(k["top_c"]))
    full = _llr(k, True, None)
    assert np.all(np.abs(partial - k["expected_llr"]) > 0.05)
    assert np.all(np.abs(full - k["expected_llr"]) > 0.01)


def test_kat1_identical_models_score_zero(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))
    world = orc.Gmm(k["w"], k["mean_world"], k["covinv"])
    x = k["x"].astype(np.float64)[:26]
    d = orc.llk_determine_top(world, x, 10, True)
    lc = orc.llk_use_top(world, x, d["idx"], d["nontop_lk"], True)
    assert abs(lc.mean() - d["llk"].mean()) < 1e-12      # test1.validate.res:2  (-5.5e-16)


def test_kat2_traintarget_map_means(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat2_traintarget.npz"))
    world = orc.Gmm(k["w"], k["mean_world"], k["covinv"])
    x = k["x"].astype(np.float64)
    sel = np.concatenate([x[b:b + n] for b, n in zip(k["seg_begin"], k["seg_len"])])
    acc = orc.em_accumulate(world, sel)
    assert acc["count"] == len(sel) == 32
    w_ml, mean_ml, _ = orc.em_get(acc, k["mean_world"], 1.0 / k["covinv"])
    got = orc.map_occdep_mean(k["mean_world"], w_ml, mean_ml, acc["count"], float(k["reg_factor"]))
    diff = np.abs(got - k["mean_expected"])
    assert np.median(diff) < float(k["median_tol"])
    assert diff.max() < float(k["max_tol"])              # single damaged mantissa bytes
    # exclusive-end frame selection would miss by >0.1
    sel2 = np.concatenate([x[b:b + n - 1] for b, n in zip(k["seg_begin"], k["seg_len"])])
    acc2 = orc.em_accumulate(world, sel2)
    w2, m2, _ = orc.em_get(acc2, k["mean_world"], 1.0 / k["covinv"])
    bad = orc.map_occdep_mean(k["mean_world"], w2, m2, acc2["count"], float(k["reg_factor"]))
    assert np.abs(bad - k["mean_expected"]).max() > 0.05


def test_kat4_normfeat_frameacc(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat4_normfeat.npz"))
    x = k["x"].astype(np.float64)
    rows = np.concatenate([np.arange(b, b + n) for b, n in zip(k["seg_begin"], k["seg_len"])])
    s, ss, n = orc.frame_acc(x[rows])
    mean, cov = orc.frame_mean_cov(s, ss, n)
    got = (x[rows] - mean) / np.sqrt(cov)                 # biased std (ddof=0)
    diff = np.abs(got - k["x_norm"][rows].astype(np.float64))
    assert np.median(diff) < float(k["median_tol"])
    assert diff.max() < float(k["max_tol"])
    unbiased = (x[rows] - mean) / np.sqrt(cov * n / (n - 1))
    assert np.median(np.abs(unbiased - k["x_norm"][rows])) > 1e-3


def test_bagging_is_seeded_and_chunked():
    # TrainTools.cpp:1070 seed, GeneralTools.cpp:455-510 chunking into 3..7-frame pieces
    b, l, s = orc.bagged_segments(221, [0, 100], [50, 25], 1.0, 3, 7)
    assert l.sum() == 75 and l.max() <= 7
    assert np.all(b[s == 1] >= 100)
    b2, l2, _ = orc.bagged_segments(221, [0, 100], [50, 25], 0.5)
    b3, l3, _ = orc.bagged_segments(221, [0, 100], [50, 25], 0.5)
    assert np.array_equal(b2, b3) and np.array_equal(l2, l3) and 0 < l2.sum() < 75


# ---- KAT-5: the GmmTokenizer goldens (LIA_Utils/GmmTokenizer/test/) -- INTEGER pins of DETERMINE_TOP_DISTRIBS -------------
def collapse_runs(v):
    """consecutive repeats removed (what test1.sym.ref holds; see the fixture's note)"""
    v = np.asarray(v)
    return v[np.r_[True, v[1:] != v[:-1]]] if len(v) else v


def confusion_counts(idx, nbest, C):
    """computeConfusionMatrix (GmmTokenizer.cpp:69-76): mce(v[0].idx, v[i].idx)++ for i < nBest, per frame"""
    M = np.zeros((C, C), np.int64)
    for row in np.asarray(idx):
        for j in range(nbest):
            M[row[0], row[j]] += 1
    return M


def kat5_selected(k):
    return np.concatenate([np.arange(b, b + n) for b, n in zip(k["seg_begin"], k["seg_len"])])


def test_kat5_gmmtokenizer_symbols_and_confusion_matrix(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat5_gmmtokenizer.npz"))
    world = orc.Gmm(k["w"], k["mean"], k["covinv"])
    x = k["x"].astype(np.float64)[kat5_selected(k)]
    nbest = int(k["nbest"])
    assert len(x) == 37 and nbest == 20
    d = orc.llk_determine_top(world, x, nbest, True)
    assert np.array_equal(collapse_runs(d["idx"][:, 0]), k["symbols"])                     # test1.sym.ref, 9 integers
    M = confusion_counts(d["idx"], nbest, 128)
    assert np.array_equal(M, k["confusion"]) and M.sum() == 37 * 20                       # mce_matrix.mat.ref, 16 384 integers
    # the top-1 stream does not depend on the list length; the matrix does and discriminates it
    for ctop in (1, 6, 10):
        dc = orc.llk_determine_top(world, x, ctop, True)
        assert np.array_equal(dc["idx"], d["idx"][:, :ctop])
    M6 = confusion_counts(d["idx"], 6, 128)                                                # the cfg's stale topDistribsCount 6
    assert M6.sum() == 222 and not np.array_equal(M6, k["confusion"])
    assert not np.array_equal(confusion_counts(d["idx"], 19, 128), k["confusion"])
    d21 = orc.llk_determine_top(world, x, 21, True)
    assert not np.array_equal(confusion_counts(d21["idx"], 21, 128), k["confusion"])
    # exclusive-end frame selection (36 or 35 frames) cannot give 740 counts
    assert k["confusion"].sum() % 20 == 0 and k["confusion"].sum() // 20 == 37
    # how sharp the pin is: the closest call between the 20th and the 21st Gaussian over the 37 frames (log domain)
    gap = np.log(d21["lk"][:, 19]) - np.log(d21["lk"][:, 20])
    assert gap.min() > 1e-6


# ---- KAT-6 (ASSUMED, not a pin): EnergyDetector's golden label file -- the variance path of getEM + varianceControl ------------
def energy_select_frames(energy, threshold, seg_begin, seg_len):
    """selectFrames (EnergyDetector.cpp:118-157) as written: a run that reaches the end of an input segment is one frame longer"""
    out, ind, begin, inside = [], 0, 0, False
    for b, n in zip(seg_begin, seg_len):
        for t in range(int(b), int(b) + int(n)):
            if energy[t] > threshold:
                if not inside:
                    inside, begin = True, ind
            elif inside:
                inside = False
                out.append((begin, ind - begin))
            ind += 1
        if inside:
            inside = False
            out.append((begin, ind - begin + 1))
    return out


def energy_detector_steps(k, energy, em_step):
    """EnergyDetector.cpp:227-274 with the EM + variance control step supplied by the caller (oracle here, HIP in tests/test_gpu_kat5.py):
    em_step(w, mean[C,1], cov[C,1], x[T,1] float32, global_cov[1]) -> (w, mean, cov)"""
    rows = np.concatenate([np.arange(b, b + n) for b, n in zip(k["seg_begin"], k["seg_len"])])
    x = np.ascontiguousarray(np.asarray(energy, np.float32)[rows, None])
    s, ss, n = orc.frame_acc(x.astype(np.float64))
    _, gcov = orc.frame_mean_cov(s, ss, n)
    w, mean, cov = k["init_w"].copy(), k["init_mean"][:, None].copy(), k["init_cov"][:, None].copy()
    for _ in range(int(k["nb_train_it"])):
        w, mean, cov = em_step(w, mean, cov, x, gcov)
    hi = int(np.argmax(mean[:, 0]))
    th = mean[hi, 0] - float(k["alpha"]) * np.sqrt(cov[hi, 0])
    return w, mean, cov, th


def kat6_oracle_step(k):
    def step(w, mean, cov, x, gcov):
        acc = orc.em_accumulate(orc.Gmm(w, mean, 1.0 / cov), x.astype(np.float64))
        w, mean, cov = orc.em_get(acc, mean, cov)
        cov, _, _ = orc.variance_control(cov, float(k["variance_flooring"]), float(k["variance_ceiling"]), gcov)
        return w, mean, cov
    return step


def kat6_normalised(k, over_file=False):
    e = k["energy"].astype(np.float64)
    sel = e if over_file else e[int(k["seg_begin"][0]):int(k["seg_begin"][0]) + int(k["seg_len"][0])]
    return ((e - sel.mean()) / sel.std()).astype(np.float32)          # biased std, like NormFeat (KAT-4)


def test_kat6_energydetector_assumed(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat6_energydetector_assumed.npz"))
    for over_file in (False, True):                                   # either normalisation window gives the golden
        e = kat6_normalised(k, over_file)
        w, mean, cov, th = energy_detector_steps(k, e, kat6_oracle_step(k))
        assert np.array_equal(np.nonzero(e[:26] > th)[0], k["expected_frames"])
        segs = energy_select_frames(e, th, k["seg_begin"], k["seg_len"])
        assert segs == [tuple(k["expected_seg"])]
        b, n = segs[0]
        assert "%g %g speech" % (b * 0.01, (b + n - 1) * 0.01) == str(k["expected_label"])
    assert abs(w.sum() - 1.0) < 1e-12
    # what the coarse output does discriminate: the VARIANCES must be re-estimated (a means / weights-only EM selects 20..25), at least
    # three iterations must run (one or two select 20..25), alpha 0.25 (0 -> 22..25, 1 -> 20..25).  It does not discriminate the floor
    # (0 .. 0.9 x globalCov all give 21..25) although both variances end ON the floor 0.5 x globalCov.
    e = kat6_normalised(k)
    w, mean, cov, th = energy_detector_steps(k, e, kat6_oracle_step(k))
    assert cov[0, 0] == cov[1, 0] and abs(cov[0, 0] - 0.5) < 1e-6      # 0.5 x globalCov of the float32 column

    def frozen_variances(w, mean, cov, x, gcov):
        acc = orc.em_accumulate(orc.Gmm(w, mean, 1.0 / cov), x.astype(np.float64))
        w2, m2, _ = orc.em_get(acc, mean, cov)
        return w2, m2, cov
    *_, th2 = energy_detector_steps(k, e, frozen_variances)
    assert np.array_equal(np.nonzero(e[:26] > th2)[0], np.arange(20, 26))
    k2 = dict(k); k2["nb_train_it"] = np.array(2)
    *_, th3 = energy_detector_steps(k2, e, kat6_oracle_step(k2))
    assert np.array_equal(np.nonzero(e[:26] > th3)[0], np.arange(20, 26))
    # the raw column does NOT give the golden (the assumption this fixture rests on)
    raw = k["energy"]
    *_, thr = energy_detector_steps(k, raw, kat6_oracle_step(k))
    assert np.array_equal(np.nonzero(raw[:26] > thr)[0], k["raw_frames"])
