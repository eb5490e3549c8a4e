"""Pins the CPU oracle against the reference's own known-answer fixtures.

KAT-1  LIA_SpkDet/ComputeTest/test/test1.validate.res  (LLK, top-10, COMPLETE, inclusive end)
KAT-2  LIA_SpkDet/TrainTarget/test/test1.validate.gmm  (full-posterior E-step + MAPOccDep means)
KAT-4  LIA_SpkDet/NormFeat/test/test1.validate.prm     (FrameAccGD mean / biased std)
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
