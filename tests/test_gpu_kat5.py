"""KAT-5 on the GPU: the reference's GmmTokenizer goldens (LIA_Utils/GmmTokenizer/test/test1.sym.ref, mce_matrix.mat.ref) -- the only
reference-held INTEGER outputs of DETERMINE_TOP_DISTRIBS + getTopDistribIndexVector (GmmTokenizer.cpp:69-76, :99-104) -- through every
top-C selection path of the C ABI.  Bit-exact: np.array_equal on the 9 symbols and on all 16 384 cells of the confusion matrix.
The matrix pins the top-1 index and the MEMBERSHIP of the top-20 set of each of the 37 selected frames (not the order of ranks 2..20)."""
import gzip
import os

import numpy as np
import pytest

from test_oracle_kat import collapse_runs, confusion_counts, kat5_selected

pytestmark = pytest.mark.gpu

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_files")


@pytest.fixture(scope="module")
def ctx():
    from lia_ral_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def k5(golden_dir):
    return np.load(os.path.join(golden_dir, "kat5_gmmtokenizer.npz"))


# (options, ctop): which kernel family serves the call is stated next to each
PATHS = [
    ({}, 20),                                    # default at ctop 20: stored likelihoods + k_topc_from_z (topc_z.hip)
    ({"topc_z": 0}, 20),                         # LDS selection kernel k_topc_determine (direct-form logits)
    ({"short_calls": 0}, 20),                    # the long-call kernel shapes on a 37-frame call
    ({}, 100),                                   # ctop > 64: any-shape k_topc_determine_big; its first 20 entries are the top-20
    ({}, 128),                                   # the whole model, sorted
    ({}, 64), ({}, 60), ({}, 21),
]
TOP1_PATHS = [
    ({}, 1), ({}, 6), ({}, 10), ({}, 16),        # fused: candidates from the epilogue of k_llk_mfma<TC>, ranked by k_topc_rank2
    ({"topc_rank2": 0}, 10),                     # ... ranked by k_topc_rank
    ({"topc_rank_direct": 1}, 10),
    ({"topc_fused": 0}, 10),                     # stored-likelihood path at ctop <= 16
    ({"topc_fused": 0, "topc_z": 0}, 10),        # LDS selection kernel at ctop <= 16
    ({"glds": 0}, 16),
]


def with_options(ctx, opts, fn):
    prev = {k: ctx.set_option(k, v) for k, v in opts.items()}
    try:
        return fn()
    finally:
        for k, v in prev.items():
            ctx.set_option(k, v)


@pytest.mark.parametrize("opts,ctop", PATHS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_kat5_confusion_matrix_and_symbols(ctx, k5, opts, ctop, dtype):
    g = ctx.gmm(k5["w"], k5["mean"], k5["covinv"])
    x = np.ascontiguousarray(k5["x"][kat5_selected(k5)].astype(dtype))
    d = with_options(ctx, opts, lambda: g.llk_determine_top(x, ctop, True))
    assert d["idx"].shape == (37, ctop)
    assert np.array_equal(collapse_runs(d["idx"][:, 0]), k5["symbols"])
    assert np.array_equal(confusion_counts(d["idx"], 20, 128), k5["confusion"])
    assert not np.array_equal(confusion_counts(d["idx"], 19, 128), k5["confusion"])


@pytest.mark.parametrize("opts,ctop", TOP1_PATHS)
def test_kat5_symbols_through_the_short_list_paths(ctx, k5, opts, ctop):
    g = ctx.gmm(k5["w"], k5["mean"], k5["covinv"])
    x = np.ascontiguousarray(k5["x"][kat5_selected(k5)])
    d = with_options(ctx, opts, lambda: g.llk_determine_top(x, ctop, True))
    assert np.array_equal(collapse_runs(d["idx"][:, 0]), k5["symbols"])
    # every short list is the head of the golden's top-20 sets: its confusion counts are dominated cell by cell by the golden
    # matrix, and row sums are ctop x (frames won by that Gaussian)
    M = confusion_counts(d["idx"], ctop, 128)
    assert np.all(M <= k5["confusion"]) and M.sum() == 37 * ctop
    assert np.array_equal(M.sum(axis=1) * 20, k5["confusion"].sum(axis=1) * ctop)


def test_kat5_per_segment_calls_and_device_pointers(ctx, k5):
    """the reference evaluates frame by frame; any batching of the 37 frames must give the same integers: per segment, per frame, and the
    whole file with the unselected frames in between (device-resident input)"""
    import torch
    g = ctx.gmm(k5["w"], k5["mean"], k5["covinv"])
    rows = kat5_selected(k5)
    per_seg = np.concatenate([g.llk_determine_top(np.ascontiguousarray(k5["x"][b:b + n]), 20, True)["idx"]
                              for b, n in zip(k5["seg_begin"], k5["seg_len"])])
    assert np.array_equal(confusion_counts(per_seg, 20, 128), k5["confusion"])
    per_frame = np.concatenate([g.llk_determine_top(np.ascontiguousarray(k5["x"][t:t + 1]), 20, True)["idx"] for t in rows])
    assert np.array_equal(per_frame, per_seg)
    xd = torch.from_numpy(np.ascontiguousarray(k5["x"])).cuda()
    whole = g.llk_determine_top(xd, 20, True)["idx"]
    whole = whole.cpu().numpy() if hasattr(whole, "cpu") else np.asarray(whole)
    assert np.array_equal(whole[rows], per_seg)


def test_kat5_topgauss_compute(ctx, k5, tmp_path):
    """gmmiv_topgauss_compute (TopGauss::compute, TopGauss.cpp:136-198) with topGauss = 20 over a list of cap entries: the stored
    selection is the golden's top-20 set"""
    from lia_ral_amd import host_capi as h
    rows = kat5_selected(k5)
    x = np.ascontiguousarray(k5["x"])
    ubm = (k5["w"], k5["mean"], 1.0 / k5["covinv"])
    for cap in (20, 64, 128):
        tg = h.topgauss(x, k5["seg_begin"], k5["seg_len"], ubm, 20.0, str(tmp_path / ("tg%d" % cap)), top_distribs_count=cap)
        nbg, idx = np.asarray(tg["nbg"]), np.asarray(tg["idx"])
        assert len(nbg) == 37 and np.all(nbg == 20)
        sel = idx.reshape(37, 20)
        assert np.array_equal(collapse_runs(sel[:, 0]), k5["symbols"])
        assert np.array_equal(confusion_counts(sel, 20, 128), k5["confusion"])


def test_kat5_gmmtokenizer_from_the_reference_files(tmp_path, k5):
    """End to end from the reference's own files: the RAW world model `wld` (intact: 68 744 bytes) read by host/io.cpp, test1.prm with
    featureServerMask 0-15,17-32, test1.lbl / label male -> computeSymbols and computeConfusionMatrix of the host layer; the matrix
    file written in DT format holds the reference's mce_matrix.mat.ref token for token (the shipped file has no blank before each newline
    while ComputeTest/test/zero.mat -- the DT file the writer is pinned to byte for byte -- has one: whitespace is not compared)."""
    from lia_ral_amd import host_capi as h
    wld = os.path.join(REF, "gmmtokenizer_wld.raw.gmm")
    prm = os.path.join(REF, "test1.prm")
    lbl = os.path.join(REF, "computetest_test1.lbl")          # GmmTokenizer/test/test1.lbl is the same file (md5)
    out = str(tmp_path / "mce_matrix.mat")
    sym, conf = h.gmm_tokenizer_files(wld, prm, lbl, mask="0-15,17-32", label="male", top_c=20, matrix_path=out)
    assert len(sym) == 37
    golden_sym = np.array(open(os.path.join(REF, "gmmtokenizer_test1.sym.ref")).read().split(), dtype=np.int64)
    assert np.array_equal(collapse_runs(sym), golden_sym) and np.array_equal(golden_sym, k5["symbols"])
    assert np.array_equal(conf, k5["confusion"])
    golden_bytes = gzip.open(os.path.join(REF, "gmmtokenizer_mce_matrix.mat.ref.gz")).read()
    assert open(out, "rb").read().split() == golden_bytes.split() and len(golden_bytes.split()) == 2 + 128 * 128
    # the cfg's (stale) topDistribsCount 6 gives another matrix: the fixture discriminates the list length
    _, conf6 = h.gmm_tokenizer_files(wld, prm, lbl, mask="0-15,17-32", label="male", top_c=6)
    assert conf6.sum() == 222 and not np.array_equal(conf6, k5["confusion"])


# ---- KAT-6 (ASSUMED, not a pin): EnergyDetector -- 10 x (full EM with variances, varianceControl) on a 2-Gaussian, 1-dimensional model ----
def test_kat6_energydetector_assumed_on_gpu(ctx, golden_dir):
    """C = 2, D = 1 is also an edge shape: one padded MFMA tile, one padded k-step.  Every iteration through gmmiv_em_accumulate ->
    gmmiv_em_get -> gmmiv_variance_control; the model after 10 iterations agrees with the oracle's to 1e-12, the frames and the label
    line are the reference's test1.validate.enr.lbl (under the fixture's normalisation assumption)."""
    from test_oracle_kat import (energy_detector_steps, energy_select_frames, kat6_normalised, kat6_oracle_step)
    k = np.load(os.path.join(golden_dir, "kat6_energydetector_assumed.npz"))
    e = kat6_normalised(k)

    def hip_step(w, mean, cov, x, gcov):
        g = ctx.gmm(w, mean, 1.0 / cov)
        acc = g.em_accumulate(x)
        w2, m2, c2 = g.em_get(acc, mean, cov)
        c2 = ctx.variance_control(c2, float(k["variance_flooring"]), float(k["variance_ceiling"]), gcov, 2, 1)
        c2 = c2[0] if isinstance(c2, tuple) else c2
        return w2, m2, np.asarray(c2).reshape(2, 1)

    w, mean, cov, th = energy_detector_steps(k, e, hip_step)
    wo, mo, co, tho = energy_detector_steps(k, e, kat6_oracle_step(k))
    assert np.max(np.abs(w - wo)) < 1e-12 and np.max(np.abs(mean - mo)) < 1e-12 and np.max(np.abs(cov - co)) < 1e-12 and abs(th - tho) < 1e-12
    assert np.array_equal(np.nonzero(e[:26] > th)[0], k["expected_frames"])
    assert energy_select_frames(e, th, k["seg_begin"], k["seg_len"]) == [tuple(k["expected_seg"])]


def test_kat6_energydetector_through_the_host_layer(golden_dir):
    from lia_ral_amd import host_capi as h
    from test_oracle_kat import energy_detector_steps, kat6_normalised, kat6_oracle_step
    k = np.load(os.path.join(golden_dir, "kat6_energydetector_assumed.npz"))
    e = kat6_normalised(k)
    r = h.energy_detector(e, k["seg_begin"], k["seg_len"], C=2, nb_train_it=int(k["nb_train_it"]), variance_flooring=float(k["variance_flooring"]),
                          variance_ceiling=float(k["variance_ceiling"]), alpha=float(k["alpha"]))
    assert list(r["begin"]) == [21] and list(r["length"]) == [6]
    b, n = int(r["begin"][0]), int(r["length"][0])
    assert "%g %g speech" % (b * 0.01, (b + n - 1) * 0.01) == str(k["expected_label"])          # "0.21 0.26 speech"
    wo, mo, co, tho = energy_detector_steps(k, e, kat6_oracle_step(k))
    assert np.max(np.abs(r["w"] - wo)) < 1e-12 and np.max(np.abs(r["mean"] - mo[:, 0])) < 1e-12 and np.max(np.abs(r["cov"] - co[:, 0])) < 1e-12
    assert abs(r["threshold"] - tho) < 1e-12
    # the raw column: another answer (what the assumption is about), and a three-Gaussian model runs too
    raw = h.energy_detector(k["energy"], k["seg_begin"], k["seg_len"], C=2, alpha=float(k["alpha"]))
    assert not (list(raw["begin"]) == [21] and list(raw["length"]) == [6])
    r3 = h.energy_detector(e, k["seg_begin"], k["seg_len"], C=3, alpha=float(k["alpha"]))
    assert abs(r3["w"].sum() - 1.0) < 1e-12 and np.all(r3["cov"] > 0)


def test_c99_example_reproduces_the_reference_llr(tmp_path, golden_dir):
    """examples/computetest_llr.c -- a C99 program that sees nothing but include/gmmiv.h and libgmmiv.so -- on the ComputeTest golden (KAT-1:
    world `wld`, client `test1`, frames 0..25 of test1.prm): prints the reference's 5.06601 (fixture tolerance 5e-5)."""
    import re
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "computetest_llr")
    subprocess.run(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "computetest_llr.c"), "-L",
                    os.path.join(root, "lia_ral_amd", "csrc"), "-lgmmiv", "-Wl,-rpath," + os.path.join(root, "lia_ral_amd", "csrc"), "-Wl,-rpath,/opt/rocm/lib",
                    "-lm", "-o", exe], check=True)
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))
    for seg, want in zip(range(2), k["expected_llr"]):
        b, n = int(k["seg_begin"][seg]), int(k["seg_len"][seg])
        x = np.ascontiguousarray(k["x"][b:b + n], np.float32)
        C, D = k["mean_world"].shape
        path = str(tmp_path / ("model%d.bin" % seg))
        with open(path, "wb") as f:
            f.write(struct.pack("<4i", C, D, n, int(k["top_c"])))
            for a in (k["w"], k["mean_world"], k["covinv"], k["mean_client"]):
                f.write(np.ascontiguousarray(a, "<f8").tobytes())
            f.write(x.tobytes())
        out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        llr = float(re.search(r"LLR (-?[0-9.]+)", out.stdout).group(1))
        assert abs(llr - float(want)) < float(k["abs_tol"]) + 1e-6, (out.stdout, want)
