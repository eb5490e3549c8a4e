#!/usr/bin/env python3
"""Regenerate the known-answer fixtures under tests/golden/ from the reference's own
test DATA files (run in the build container only; /root/reference is absent on the GPU box).

What is extracted (data only -- model parameters, feature frames, expected numbers):
  KAT-1  LIA_SpkDet/ComputeTest/test/{wld,test1,test1.prm,test1.lbl,test1.validate.res}
  KAT-2  LIA_SpkDet/TrainTarget/test/{wld,test1.prm,test1.lbl,test1.validate.gmm}
  KAT-4  LIA_SpkDet/NormFeat/test/{test1.prm,test1.validate.prm}
  KAT-5  LIA_Utils/GmmTokenizer/test/{wld,test1.prm,test1.lbl,test1.sym.ref,mce_matrix.mat.ref}
         (integers: the per-frame best Gaussian and the top-20 confusion counts of DETERMINE_TOP_DISTRIBS)
  KAT-6  LIA_SpkDet/EnergyDetector/test/{test1.prm,test1.lbl,test1.validate.enr.lbl}   -- ASSUMED, not a pin (see the note it carries)

The binary files in the reference checkout went through text-mode newline translation
(every 0x0D 0x0A lost its 0x0D; lone 0x0D became 0x0A).  RAW GMM files are
repaired here by re-inserting the dropped 0x0D bytes where the per-Gaussian redundancy
(cst, det vs covInv) says a byte is missing.  Lone 0x0D->0x0A substitutions cannot be
recovered; they perturb single mantissa bytes (tolerances are stored with the fixtures).
"""
import os, struct, sys
import numpy as np

REF = "/root/reference/LIA_SpkDet"
OUT = os.path.dirname(os.path.abspath(__file__))


def gauss_ok(buf, off, D):
    """Consistency of one RAW Gaussian record at byte offset off: cst,det,flag,covInv[D],mean[D]."""
    need = 17 + 16 * D
    if off + need > len(buf):
        return False
    cst, det = struct.unpack_from("<dd", buf, off)
    civ = np.frombuffer(buf, "<f8", D, off + 17)
    mu = np.frombuffer(buf, "<f8", D, off + 17 + 8 * D)
    if not (np.all(np.isfinite(civ)) and np.all(civ > 0) and np.all(np.isfinite(mu))):
        return False
    if np.max(np.abs(mu)) > 20 or not np.isfinite(det) or det <= 0:
        return False
    det2 = float(np.prod(1.0 / civ))
    cst2 = (2 * np.pi) ** (-D / 2.0) / np.sqrt(det2)
    return abs(det2 - det) <= 0.02 * det and abs(cst2 - cst) <= 0.02 * abs(cst)


def plausible(buf, off, D):
    """Weaker test for records that also carry substituted bytes: finite, sane magnitudes."""
    if off + 17 + 16 * D > len(buf):
        return False
    civ = np.frombuffer(buf, "<f8", D, off + 17)
    mu = np.frombuffer(buf, "<f8", D, off + 17 + 8 * D)
    with np.errstate(all="ignore"):
        return bool(np.all(np.isfinite(civ)) and np.all(np.isfinite(mu)) and np.all(civ > 0.05)
                    and np.all(civ < 1e3) and np.max(np.abs(mu)) < 20)


def repair_raw_gmm(raw):
    """Return (bytes, n_inserted, damaged) with dropped 0x0D bytes re-inserted.

    Walk the Gaussian records in order.  A record that fails the cst/det redundancy check while
    the FOLLOWING records pass at their nominal offsets only carries substituted bytes (kept,
    listed in `damaged`).  Otherwise a byte is missing at or before it: try a 0x0D in front of
    every 0x0A of the previous+current record and keep the first that makes this record and
    the next two consistent.
    """
    buf = bytearray(raw)
    C, D = struct.unpack_from("<II", buf, 0)
    rec = 17 + 16 * D
    base = 8 + 8 * C
    want = base + C * rec
    ins, damaged = 0, []

    def ok(b, g):
        if g >= C:
            return len(b) - base >= C * rec - (want - len(b))  # past the end: no evidence
        return gauss_ok(b, base + g * rec, D)

    g = 0
    while g < C:
        if ok(buf, g):
            g += 1
            continue
        if want - len(buf) == 0 or (ok(buf, g + 1) and ok(buf, g + 2)):
            damaged.append(g)          # substitution damage only, alignment intact
            g += 1
            continue
        off = base + g * rec
        lo = max(8, off - rec) if g > 0 else 8
        hi = min(len(buf), off + rec)
        best, best_score = None, -1.0
        for p in range(lo, hi + rec):
            if p >= len(buf) or buf[p] != 0x0A:
                continue
            cand = buf[:p] + b"\x0d" + buf[p:]
            if not (ok(cand, g + 2) and ok(cand, g + 3)):
                continue                      # alignment must be restored two records on
            score = sum(ok(cand, k) for k in range(max(0, g - 1), g + 2))
            score += 0.1 * sum(plausible(cand, base + k * rec, D) for k in range(max(0, g - 1), g + 2))
            if score > best_score:
                best, best_score = cand, score
        if best is None:
            raise RuntimeError("cannot repair Gaussian %d" % g)
        buf = best
        ins += 1
        if not ok(buf, g):
            damaged.append(g)
        g += 1
    if len(buf) != want:
        raise RuntimeError("repair incomplete: %d != %d" % (len(buf), want))
    return bytes(buf), ins, damaged


def parse_raw_gmm(b):
    C, D = struct.unpack_from("<II", b, 0)
    w = np.frombuffer(b, "<f8", C, 8).copy()
    rec = 17 + 16 * D
    cst = np.empty(C); det = np.empty(C)
    civ = np.empty((C, D)); mu = np.empty((C, D))
    for g in range(C):
        off = 8 + 8 * C + g * rec
        cst[g], det[g] = struct.unpack_from("<dd", b, off)
        civ[g] = np.frombuffer(b, "<f8", D, off + 17)
        mu[g] = np.frombuffer(b, "<f8", D, off + 17 + 8 * D)
    return dict(w=w, cst=cst, det=det, covinv=civ, mean=mu)


def load_gmm(path):
    raw = open(path, "rb").read()
    fixed, n, damaged = repair_raw_gmm(raw)
    m = parse_raw_gmm(fixed)
    m["n_inserted"] = n
    m["damaged"] = damaged
    return m


def load_prm(path):
    b = open(path, "rb").read()
    hdr = struct.unpack_from("<4I", b, 0)      # (2, base dim, nframes, flags)
    dim = (len(b) - 16) // (4 * hdr[2])
    x = np.frombuffer(b, "<f4", hdr[2] * dim, 16).reshape(hdr[2], dim).copy()
    return hdr, x


def mask_0_15_17_32(x):
    idx = list(range(0, 16)) + list(range(17, 33))
    return x[:, idx]


def main():
    # ---- KAT-1 ----
    d = REF + "/ComputeTest/test"
    wld = load_gmm(d + "/wld")
    cli = load_gmm(d + "/test1")
    _, x = load_prm(d + "/test1.prm")
    assert np.array_equal(wld["w"], cli["w"])
    np.savez_compressed(
        OUT + "/kat1_computetest.npz",
        w=wld["w"], covinv=wld["covinv"], mean_world=wld["mean"],
        covinv_client=cli["covinv"], mean_client=cli["mean"], w_client=cli["w"],
        x=mask_0_15_17_32(x).astype(np.float32),
        seg_begin=np.array([0, 30]), seg_len=np.array([26, 11]),       # test1.lbl, inclusive end
        expected_llr=np.array([5.06601, 4.26793]),                     # test1.validate.res:1,3
        top_c=np.array(10), min_llk=np.array(-200.0), max_llk=np.array(200.0),
        abs_tol=np.array(5e-5),
    )
    print("KAT-1: inserted", wld["n_inserted"], cli["n_inserted"])
    # ---- KAT-2 ----
    d = REF + "/TrainTarget/test"
    wld2 = load_gmm(d + "/wld")
    val = load_gmm(d + "/test1.validate.gmm")
    _, x2 = load_prm(d + "/test1.prm")
    np.savez_compressed(
        OUT + "/kat2_traintarget.npz",
        w=wld2["w"], covinv=wld2["covinv"], mean_world=wld2["mean"],
        mean_expected=val["mean"], w_expected=val["w"], covinv_expected=val["covinv"],
        x=mask_0_15_17_32(x2).astype(np.float32),
        seg_begin=np.array([0, 20]), seg_len=np.array([11, 21]),       # TrainTarget/test/test1.lbl
        reg_factor=np.array(10.0), median_tol=np.array(1e-7), max_tol=np.array(2e-3),
    )
    print("KAT-2: inserted", wld2["n_inserted"], val["n_inserted"])
    # ---- KAT-4 ----
    d = REF + "/NormFeat/test"
    _, xin = load_prm(d + "/test1.prm")
    _, xout = load_prm(d + "/test1.validate.prm")
    np.savez_compressed(
        OUT + "/kat4_normfeat.npz", x=xin, x_norm=xout,
        seg_begin=np.array([0, 30]), seg_len=np.array([11, 11]),
        median_tol=np.array(1e-6), max_tol=np.array(0.05),
    )
    print("KAT-4 ok")
    # ---- KAT-5: GmmTokenizer (LIA_Utils/GmmTokenizer/src/GmmTokenizer.cpp:69-76 confusion matrix, :99-104 symbols) ----
    # wld is intact here (68 744 bytes = 8 + 8*128 + 128*(17 + 16*32): nothing dropped, repair inserts nothing); one record (81)
    # fails the cst/det redundancy check -- a substituted low mantissa byte in cst/det, which are recomputed and never used.
    d = "/root/reference/LIA_Utils/GmmTokenizer/test"
    raw = open(d + "/wld", "rb").read()
    fixed, n_ins, damaged = repair_raw_gmm(raw)
    assert fixed == raw and n_ins == 0
    tok = parse_raw_gmm(raw)
    _, x5 = load_prm(d + "/test1.prm")
    sym = np.array(open(d + "/test1.sym.ref").read().split(), dtype=np.int64)
    lines = open(d + "/mce_matrix.mat.ref").read().split("\n")
    assert lines[0].split() == ["128", "128"]                              # DT matrix: "rows cols" then one row per line
    conf = np.array([[int(float(v)) for v in ln.split()] for ln in lines[1:129]], dtype=np.int64)
    assert conf.shape == (128, 128)
    np.savez_compressed(
        OUT + "/kat5_gmmtokenizer.npz",
        w=tok["w"], covinv=tok["covinv"], mean=tok["mean"],
        x=mask_0_15_17_32(x5).astype(np.float32),
        seg_begin=np.array([0, 30]), seg_len=np.array([26, 11]),           # test1.lbl "0 0.25 male" / "0.3 0.4 male", inclusive end
        symbols=sym,                                                       # test1.sym.ref: 9 values for 37 selected frames
        confusion=conf,                                                    # mce_matrix.mat.ref: sum 740 = 37 frames x 20
        nbest=np.array(20),
        note=np.array(
            "nbest: GmmTokenizer.cfg says topDistribsCount 6, but the shipped matrix sums to 740 = 37 x 20 (6 would give 222): the "
            "golden was produced with 20, the cfg is stale (like KAT-2's nbTrainIt).  symbols: computeSymbols (GmmTokenizer.cpp:99-104) "
            "adds ONE value per selected frame (37); test1.sym.ref holds 9 -- it equals the per-frame best Gaussian with consecutive "
            "repeats collapsed, exactly.  ULongVector::save is alize-core (not in the tree), so whether the collapse happened in save() "
            "or in an older computeSymbols is not visible; the 9-integer match of the collapsed stream is the pin."),
    )
    import gzip, shutil
    rf = OUT + "/ref_files"
    shutil.copyfile(d + "/wld", rf + "/gmmtokenizer_wld.raw.gmm")          # reference test DATA (a model file), not source
    shutil.copyfile(d + "/test1.sym.ref", rf + "/gmmtokenizer_test1.sym.ref")
    with gzip.GzipFile(rf + "/gmmtokenizer_mce_matrix.mat.ref.gz", "wb", mtime=0) as g:
        g.write(open(d + "/mce_matrix.mat.ref", "rb").read())
    # ---- KAT-6 (assumed): EnergyDetector (LIA_SpkDet/EnergyDetector/src/EnergyDetector.cpp:163-183 fixed init, :227-263 training
    # loop + meanStd threshold, :128-157 selectFrames) -- the one reference artefact that goes through the VARIANCE estimate of getEM
    d = REF + "/EnergyDetector/test"
    _, x6 = load_prm(d + "/test1.prm")
    lab_in = open(d + "/test1.lbl").read().split()
    lab_out = open(d + "/test1.validate.enr.lbl").read().split()
    assert lab_in == ["0", "0.25", "male"] and lab_out == ["0.21", "0.26", "speech"]
    np.savez_compressed(
        OUT + "/kat6_energydetector_assumed.npz",
        energy=x6[:, 16].astype(np.float32),                               # featureServerMask 16, vectSize 1
        seg_begin=np.array([0]), seg_len=np.array([26]),                   # "0 0.25 male", inclusive end
        nb_train_it=np.array(10), variance_flooring=np.array(0.5), variance_ceiling=np.array(10.0), alpha=np.array(0.25),
        init_mean=np.array([-2.0, 2.0]), init_cov=np.array([1.0, 1.0]), init_w=np.array([0.5, 0.5]),   # energyMixtureInit, C = 2, D = 1
        expected_frames=np.arange(21, 26),                                 # frames above the threshold
        expected_seg=np.array([21, 6]),                                    # createSeg(begin, ind - begin + 1) with ind = 26 (:150-153)
        expected_label=np.array("0.21 0.26 speech"),                       # begin * 0.01, (begin + length - 1) * 0.01
        raw_frames=np.array([2, 17, 18, 19, 20, 21, 22, 23, 24, 25]),      # what the RAW energy column gives -- not the golden
        note=np.array(
            "ASSUMED, not pinned: the golden 0.21 0.26 comes out only when the energy column has mean 0 / variance 1 before the EM "
            "(the fixed init puts the two Gaussians at -2 / +2 with variance 1, and EnergyDetector.cpp:204 says the input is an energy "
            "parameter file -- in the LIA recipe NormFeat's energy normalisation runs first; the cfg does not state it and test1.prm is "
            "the shared, un-normalised file).  Normalising on the 26 selected frames or on all 50 frames of the file gives the same "
            "five frames; the raw column gives raw_frames.  The output is one coarse segment, so it constrains the variance path "
            "(getEM variances, varianceControl with flooring 0.5 x globalCov hit from iteration 3 on) only loosely."),
    )
    print("KAT-6 (assumed) ok")
    print("KAT-5: symbols", sym.tolist(), "confusion sum", int(conf.sum()), "damaged records", damaged)


if __name__ == "__main__":
    main()
