"""File formats either side of the hot path against genuine files written by
ALIZE / LIA_RAL and shipped as the reference's test data (tests/golden/ref_files/)."""
import filecmp
import os
import struct

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_files")


def test_raw_mixture_and_prm_roundtrip(tmp_path, golden_dir):
    from lia_ral_amd import host_capi as h
    raw = os.path.join(REF, "traintarget_wld.raw.gmm")
    prm = os.path.join(REF, "test1.prm")
    out_raw, out_prm = str(tmp_path / "w.gmm"), str(tmp_path / "f.prm")
    dims, mean0, frame0 = h.io_roundtrip(raw, out_raw, prm, out_prm, "0-15,17-32")
    assert tuple(dims) == (128, 32, 50, 32)
    k = np.load(os.path.join(golden_dir, "kat2_traintarget.npz"))
    assert np.array_equal(mean0, k["mean_world"][0])                # reader == fixture extraction
    assert np.array_equal(frame0, k["x"][0])                        # mask 0-15,17-32 drops column 16
    assert filecmp.cmp(prm, out_prm, shallow=False)                 # feature file re-written bit for bit
    a, b = open(raw, "rb").read(), open(out_raw, "rb").read()
    assert len(a) == len(b) and a[:8 + 8 * 128] == b[:8 + 8 * 128]  # header + weights identical
    # cst / det are recomputed (computeAll) and covInv goes through cov = 1/covInv: equal to rounding
    # (the stored cst/det may carry a substituted low mantissa byte -- SURVEY.md F3 -- hence 1e-6)
    rec = 17 + 16 * 32
    for c in (0, 17, 127):
        off = 8 + 8 * 128 + c * rec
        ca, da = struct.unpack_from("<dd", a, off); cb, db = struct.unpack_from("<dd", b, off)
        assert abs(ca - cb) <= 1e-6 * abs(ca) and abs(da - db) <= 1e-6 * abs(da)
        ia = np.frombuffer(a, "<f8", 64, off + 17); ib = np.frombuffer(b, "<f8", 64, off + 17)
        assert np.allclose(ia[:32], ib[:32], rtol=1e-15, atol=0) and np.array_equal(ia[32:], ib[32:])


def test_label_selection_inclusive_end():
    from lia_ral_amd import host_capi as h
    b, l = h.label_segments(os.path.join(REF, "computetest_test1.lbl"), "male")     # "0 0.25 male" / "0.3 0.4 male"
    assert list(b) == [0, 30] and list(l) == [26, 11]                               # SegTools.cpp:265-271
    b, l = h.label_segments(os.path.join(REF, "traintarget_test1.lbl"), "speech")   # "0 0.1" / "0.2 0.4"
    assert list(b) == [0, 20] and list(l) == [11, 21]
    b, l = h.label_segments(os.path.join(REF, "computetest_test1.lbl"), "female")
    assert len(b) == 0


def test_label_times_off_the_frame_grid(tmp_path):
    """timeToFrameIdx (SegTools.cpp:135-142) truncates time / frameLength and only moves to the next frame when the
    fractional part exceeds 0.99999 -- it does NOT round to nearest: 0.125 s -> frame 12, 5.928 s -> 592, while
    0.29 / 0.01 = 28.999999999999996 -> 29 (a whole number of frames up to rounding)."""
    from lia_ral_amd import host_capi as h
    lbl = tmp_path / "off.lbl"
    lbl.write_text("0.125 0.138 sp\n5.928 5.9399 sp\n0.07 0.29 sp\n0.0199 0.02999995 sp\n0.0199 0.0299999 sp\n")
    b, l = h.label_segments(str(lbl), "sp")
    assert list(b) == [12, 592, 7, 1, 1]
    assert list(l) == [13 - 12 + 1, 593 - 592 + 1, 29 - 7 + 1, 3 - 1 + 1, 2 - 1 + 1]   # 2.999995 -> 3, 2.99999 -> 2     # end frame inclusive (:269-270)


def test_damaged_file_is_reported(tmp_path):
    from lia_ral_amd import host_capi as h
    bad = tmp_path / "short.gmm"
    bad.write_bytes(open(os.path.join(REF, "traintarget_wld.raw.gmm"), "rb").read()[:-3])
    with pytest.raises(h.HostError, match="text-mode damaged"):
        h.io_roundtrip(str(bad), str(tmp_path / "o.gmm"), os.path.join(REF, "test1.prm"), str(tmp_path / "o.prm"), "")


def test_xml_mixture_reads_and_rewrites_the_reference_file(tmp_path):
    """LIA_SpkDet/TrainWorld/test/wld.validate (XML, 10 x 32): weights / covInv / means come back bit for bit, and the re-written
    file is the same text except for the last digits of the recomputed cst / det attributes (DistribGD::computeAll)."""
    import re
    from lia_ral_amd import host_capi as h
    src = os.path.join(REF, "trainworld_wld.validate.xml")
    out, raw = str(tmp_path / "w.xml"), str(tmp_path / "w.raw")
    dims, first = h.io_xml(src, out, raw)
    assert tuple(dims) == (10, 32)
    a, b = open(src).read().splitlines(), open(out).read().splitlines()
    assert len(a) == len(b) == 1 + 10 * (2 + 64) + 1
    assert first[0] == 0.1457922064745452162 and first[1] == 0.2845698083139924783
    num = re.compile(r'(cst|det)="([^"]+)"')
    for la, lb in zip(a, b):
        if "<DistribGD" in la:
            assert num.sub("", la) == num.sub("", lb)                     # i and weight: identical text
            for (ka, va), (kb, vb) in zip(num.findall(la), num.findall(lb)):
                assert ka == kb and abs(float(va) - float(vb)) <= 1e-12 * abs(float(va))
        else:
            assert la == lb                                               # every covInv / mean line: identical text
    # reading the re-written file gives the same text again (fixed point), and the RAW twin holds the same numbers
    out2 = str(tmp_path / "w2.xml")
    h.io_xml(out, out2)
    assert open(out).read() == open(out2).read()
    rawb = open(raw, "rb").read()
    assert struct.unpack_from("<II", rawb) == (10, 32) and np.frombuffer(rawb, "<f8", 1, 8)[0] == first[0]


def test_dt_db_matrices_and_vector_files(tmp_path):
    """DT matrix: the reference's own LIA_SpkDet/ComputeTest/test/zero.mat (32768 x 5) re-written byte for byte; DB (binary, layout
    unpinned: no DB file ships with LIA_RAL) round-trips through DT with BOTH header widths (`unsigned long` extents: 2 x 8 bytes
    from an LP64 build of ALIZE, 2 x 4 from a 32-bit / Windows one; the reader tells them apart by the file size); per-id i-vector
    files (TVAcc::saveWbyFile -> PldaTest::load)."""
    import gzip
    from lia_ral_amd import host_capi as h
    zero = tmp_path / "zero.mat"
    zero.write_bytes(gzip.open(os.path.join(REF, "computetest_zero.mat.gz")).read())
    assert h.io_db_header_bytes() == struct.calcsize("L") == 8            # default = this platform's unsigned long, like upstream built here
    for width in (8, 4):
        prev = h.io_db_header_bytes(width)
        try:
            db, dt = tmp_path / ("z%d.db" % width), tmp_path / ("z%d.dt" % width)
            assert h.io_matrix_convert(str(zero), "DT", str(db), "DB") == (32768, 5)
            assert os.path.getsize(db) == 2 * width + 8 * 32768 * 5
            raw = db.read_bytes()
            assert struct.unpack_from("<QQ" if width == 8 else "<II", raw) == (32768, 5)
            assert h.io_matrix_convert(str(db), "DB", str(dt), "DT") == (32768, 5)        # read back whatever the width
            assert filecmp.cmp(zero, dt, shallow=False)
        finally:
            h.io_db_header_bytes(prev)
    (tmp_path / "empty.db").write_bytes(b"\x00" * 16)                       # 0 x 0 with the wide header
    assert h.io_matrix_convert(str(tmp_path / "empty.db"), "DB", str(tmp_path / "empty.dt"), "DT") == (0, 0)
    rng = np.random.default_rng(0)
    W = rng.normal(size=(5, 40))
    ids = ["spk%d_a" % i for i in range(5)]
    for fmt in ("DB", "DT"):
        d = tmp_path / fmt; d.mkdir()
        back = h.io_vectors(str(d), ids, ".y", fmt, W)
        assert np.array_equal(back, W.T)                                   # one vector per COLUMN like PldaTest::_models
        assert sorted(os.listdir(d)) == sorted(i + ".y" for i in ids)
    with pytest.raises(h.HostError, match="fit neither"):
        (tmp_path / "bad.db").write_bytes(b"\x02\x00\x00\x00\x03\x00\x00\x00" + b"\x00" * 40)
        h.io_matrix_convert(str(tmp_path / "bad.db"), "DB", str(tmp_path / "o"), "DT")


def test_select_frames_of_the_energy_detector_as_written():
    """liagpu::selectFrames (EnergyDetector.cpp:118-157) against the Python restatement the KAT-6 test uses: a run that reaches the end of
    an input segment is one frame LONGER than a run that ends inside it (:144 `ind - begin` with ind = end + 1, :151 `ind - begin + 1` with
    ind already past the end) -- reproduced as written; begin counts frames of the SELECTION, not of the file."""
    from lia_ral_amd import host_capi as h
    from test_oracle_kat import energy_select_frames
    rng = np.random.default_rng(0)
    for trial in range(200):
        T = int(rng.integers(1, 60))
        e = rng.normal(size=T).astype(np.float32)
        nseg = int(rng.integers(0, 4))
        cuts = np.sort(rng.integers(0, T + 1, 2 * nseg))
        sb, sl = cuts[0::2], cuts[1::2] - cuts[0::2]
        th = float(rng.normal(0, 0.7))
        b, l, cnt = h.select_frames(e, th, sb, sl)
        ref = energy_select_frames(e, th, sb, sl)
        assert list(zip(b.tolist(), l.tolist())) == ref, (trial, e, th, sb, sl)
        assert cnt == sum(int((e[s:s + n] > th).sum()) for s, n in zip(sb, sl))
    # the asymmetry itself: one selected segment of 10 frames, the last 3 above the threshold -> length 4; an inner run of 3 -> length 3
    e = np.array([0, 0, 0, 0, 0, 0, 0, 1, 1, 1], np.float32)
    assert [x.tolist() for x in h.select_frames(e, 0.5, [0], [10])[:2]] == [[7], [4]]
    e = np.array([0, 0, 1, 1, 1, 0, 0, 0, 0, 0], np.float32)
    assert [x.tolist() for x in h.select_frames(e, 0.5, [0], [10])[:2]] == [[2], [3]]
    with pytest.raises(h.HostError, match="beyond the end"):
        h.select_frames(e, 0.5, [5], [10])
