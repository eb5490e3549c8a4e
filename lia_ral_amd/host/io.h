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

struct MatrixD { unsigned long rows = 0, cols = 0; std::vector<double> v; };
MatrixD readMatrixDT(const std::string &path);
void writeMatrixDT(const std::string &path, const MatrixD &m);

// gender, client id, decision ('1'/'0' by threshold), test file, [start end,] score
std::string resultLine(double llr, const std::string &clientName, const std::string &testName, const std::string &gender,
                       double threshold, bool withTimes = false, double start = 0.0, double end = 0.0);

} // namespace liagpu
