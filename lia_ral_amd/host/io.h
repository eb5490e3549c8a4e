// io.h -- on-disk formats either side of the hot path, decoded from the
// reference's own fixtures; host-only, little-endian.
//   * feature files (.prm as shipped under LIA_SpkDet/*/test): 16-byte header of four u32
//     (2, base dimension, frame count, flags) + frames x dim float32; `featureServerMask`
//     ("0-15,17-32") selects columns like ALIZE's FeatureServer does
//     (LIA_SpkDet/ComputeTest/test/ComputeTest.cfg:28-29)
//   * label files (.lbl): "begin_s end_s label" per line; frame = time / frameLength and the end
//     frame is INCLUSIVE (LIA_SpkTools/src/SegTools.cpp:265-271)
//   * RAW mixture files (saveMixtureFileFormat RAW): u32 C, u32 D, f64 w[C], then per Gaussian
//     f64 cst, f64 det, u8 flag, f64 covInv[D], f64 mean[D]
//   * DT matrices (text): "rows cols" then the values, row-major (ComputeTest/test/zero.mat)
//   * score lines of the NIST-style result file (LIA_SpkTools/src/IOFormat.cpp:112-122), e.g.
//     "M test1 1 test3 0 0.26 5.06601" (LIA_SpkDet/ComputeTest/test/test1.validate.res)
#pragma once
#include <string>
#include <vector>

#include "liatools_gpu.h"

namespace liagpu {

std::vector<int> parseFeatureMask(const std::string &mask);             // "0-15,17-32" -> column list
struct FeatureFile {
    unsigned long nFrames = 0, vectSize = 0; // after masking
    unsigned baseDim = 0, flags = 0;
    std::vector<float> data;                 // [nFrames x vectSize]
};
FeatureFile readFeatureFile(const std::string &path, const std::string &mask = "");
void writeFeatureFile(const std::string &path, const FeatureFile &f);   // unmasked layout, same header

struct LabelSeg { double begin_s, end_s; std::string label; };
std::vector<LabelSeg> readLabelFile(const std::string &path);
SegCluster selectSegments(const std::vector<LabelSeg> &lab, const std::string &labelSelectedFrames, double frameLength,
                          unsigned long source = 0);

MixtureGD readMixtureRAW(const std::string &path);
void writeMixtureRAW(const std::string &path, const MixtureGD &m);
// XML mixture files (saveMixtureFileFormat XML; fixture LIA_SpkDet/TrainWorld/test/wld.validate):
//   <MixtureGD version="1" id="#1" distribCount="C" vectSize="D"> / <DistribGD i weight cst det> / <covInv i>v</covInv> ... <mean i>v</mean>
// numbers carry 19 significant digits (%.19g).  weight / covInv / mean round-trip bit for bit; cst / det are recomputed from the
// covariances like DistribGD::computeAll does and agree with the file to rounding.
MixtureGD readMixtureXML(const std::string &path);
void writeMixtureXML(const std::string &path, const MixtureGD &m, const std::string &id = "#1");
MixtureGD readMixture(const std::string &path); // XML when the file starts with '<', else RAW

struct MatrixD { unsigned long rows = 0, cols = 0; std::vector<double> v; };
MatrixD readMatrixDT(const std::string &path);
void writeMatrixDT(const std::string &path, const MatrixD &m);
// DB matrices (saveMatrixFormat DB, the binary twin of DT written by alize-core's Matrix<double>::save): rows, cols, then
// rows * cols float64, little-endian.  The two extents are `unsigned long` members written with their own size: 8 bytes each
// from an LP64 build (Linux), 4 bytes each from a 32-bit / Windows build.  readMatrixDB accepts BOTH (the width is the one that
// makes the file size come out exactly); writeMatrixDB writes this platform's `unsigned long` width unless
// setMatrixDBHeaderBytes(4 | 8) says otherwise (returns the previous width).  alize-core is not part of the LIA_RAL tree and
// no DB file ships with it, so this layout is NOT pinned by a reference file; DT is (ComputeTest/test/zero.mat).
MatrixD readMatrixDB(const std::string &path);
void writeMatrixDB(const std::string &path, const MatrixD &m);
int setMatrixDBHeaderBytes(int bytes);
MatrixD readMatrix(const std::string &path, const std::string &format);   // "DT" | "DB" (loadMatrixFormat)
void writeMatrix(const std::string &path, const MatrixD &m, const std::string &format);
// Per-id vector files: TVAcc::saveWbyFile (AccumulateTVStat.cpp:2799-2822) writes row `session` of W as a 1 x rankT matrix to
// <saveVectorFilesPath><id><vectorFilesExtension>; PldaTest::load (PldaTools.cpp:3552-3588) reads <testVectorFilesPath>/<id><ext>
// back as column k of _models / _segments.
void saveVectorsById(const std::string &dir, const std::vector<std::string> &ids, const std::string &ext, const std::vector<double> &W,
                     unsigned long rank, const std::string &format = "DB");
// -> [dim x ids.size()] one vector per COLUMN (the layout of PldaTest::_models / _segments); dim from the first file
std::vector<double> loadVectorsById(const std::string &dir, const std::vector<std::string> &ids, const std::string &ext, unsigned long &dim,
                                    const std::string &format = "DB");

// gender, client id, decision ('1'/'0' by threshold), test file, [start end,] score
std::string resultLine(double llr, const std::string &clientName, const std::string &testName, const std::string &gender,
                       double threshold, bool withTimes = false, double start = 0.0, double end = 0.0);

} // namespace liagpu
