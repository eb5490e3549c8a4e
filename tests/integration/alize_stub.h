// alize_stub.h -- TEST-ONLY STUB.  alize-core (the library that defines these classes) is not part of the LIA_RAL tree and not in this
// image, so nothing here is taken from it: every declaration below is the signature that LIA_RAL's OWN call sites imply (SURVEY.md 8(b);
// the file:line next to each one is where LIA_RAL uses it that way).  It pins NOTHING about ALIZE.  Its only purpose is to let
// tests/test_cpu_integration_doc.py compile the ```cpp blocks of INTEGRATION.md against include/gmmiv.h, so that the documented
// call-site branches cannot drift from the C ABI (argument order, types, names).  Bodies are absent on purpose: -fsyntax-only.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace alize {

typedef double real_t;
class String {                                               // LIA_SpkTools/src/AccumulateStat.cpp:133 config.getParam("numThread").toLong()
  public:
    String();
    String(const char *);
    const char *c_str() const;
    long toLong() const;
    double toDouble() const;
    bool toBool() const;
    bool operator==(const char *) const;
};
class Exception {                                            // AccumulateTVStat.cpp:1947 throw Exception("...", __FILE__, __LINE__)
  public:
    Exception(const String &msg, const char *file, int line);
    String toString() const;                                 // TrainWorld.cpp:188 cout << e.toString().c_str()
};
class Config {                                               // AccumulateStat.cpp:131-140
  public:
    bool existsParam(const String &) const;
    const String &getParam(const String &) const;
};
class Feature {                                              // AccumulateTVStat.cpp:307 f.getDataVector() -> double*
  public:
    Feature();
    double *getDataVector() const;
    double operator[](unsigned long) const;
};
class FeatureServer {                                        // AccumulateStat.cpp:121-128
  public:
    unsigned long getVectSize();
    void seekFeature(unsigned long idx);
    bool readFeature(Feature &f);
    unsigned long getFirstFeatureIndexOfASource(const String &sourceName);
};
class Seg {                                                  // AccumulateStat.cpp:122-124
  public:
    unsigned long begin() const;
    unsigned long length() const;
    const String &sourceName() const;
};
class SegCluster {                                           // AccumulateStat.cpp:119-121
  public:
    void rewind();
    Seg *getSeg();
};
class DistribGD {                                            // AccumulateTVStat.cpp:154-162, TrainTools.cpp:577-582, :746-756
  public:
    double getMean(unsigned long i) const;
    double getCov(unsigned long i) const;
    double getCovInv(unsigned long i) const;
    void setMean(double v, unsigned long i);
    void setCov(double v, unsigned long i);
    void computeAll();
};
class MixtureGD {                                            // TrainTools.cpp:567-587
  public:
    unsigned long getDistribCount() const;
    unsigned long getVectSize() const;
    double &weight(unsigned long c);
    DistribGD &getDistrib(unsigned long c);
};
template <class T> class Matrix {                            // AccumulateTVStat.cpp:1960-1968 _statN.getArray(), :1969 setDimensions
  public:
    T *getArray() const;
    unsigned long rows() const;
    unsigned long cols() const;
    void setDimensions(unsigned long r, unsigned long c);
    void setAllValues(T v);
    T &operator()(unsigned long i, unsigned long j);
};
class DoubleSquareMatrix {                                   // AccumulateTVStat.cpp:1873-1890
  public:
    double *getArray() const;
    unsigned long size() const;
    void setSize(unsigned long n);
    void setAllValues(double v);
};
template <class T> class RealVector {                        // AccumulateTVStat.cpp:1966 _ubm_invvar.getArray()
  public:
    T *getArray() const;
    unsigned long size() const;
    void setSize(unsigned long n);
    void setAllValues(T v);
    T &operator[](unsigned long i);
};
typedef RealVector<double> DoubleVector;
class ULongVector {                                          // TopGauss.cpp:181 _idx.addValue(...)
  public:
    unsigned long *getArray() const;
    unsigned long size() const;
    void setSize(unsigned long n);
    void addValue(unsigned long v);
    void clear();
    unsigned long &operator[](unsigned long i);
};
class BoolMatrix {                                           // PldaTools.cpp:3871 _trials(m, s)
  public:
    bool operator()(unsigned long i, unsigned long j) const;
    unsigned long rows() const;
    unsigned long cols() const;
};

} // namespace alize

// ---- the LIA_SpkTools classes the branches live in: ONLY the members the documented branches touch, named as in
// LIA_SpkTools/include/{AccumulateTVStat.h,PldaTools.h,AccumulateJFAStat.h,TopGauss.h} (those headers ARE in the LIA_RAL tree) -------
struct gmmiv_ctx;
struct gmmiv_gmm;
struct gmmiv_comm;

class TVAcc {                                                // AccumulateTVStat.h: class TVAcc, private members
  public:
    unsigned long _vectSize, _n_distrib, _svSize, _rankT, _n_speakers, _n_sessions;
    alize::RealVector<double> _ubm_means, _ubm_invvar, _meanW;
    alize::Matrix<double> _statN, _statF, _T, _W, _A, _Cmx;
    alize::DoubleSquareMatrix _R;
    alize::DoubleVector _r;
    // added by the GPU branch
    gmmiv_ctx *_gpu;
    gmmiv_gmm *_gpuUbm;
    std::vector<double> _tettPacked, _aPacked;
    void computeAndAccumulateTVStatGpu(alize::FeatureServer &fs, std::vector<alize::SegCluster *> &segsOfLine);
    void substractMGpu();
    void estimateTETtGpu();
    void estimateWGpu();
    void estimateAandCGpu();
    void updateTestimateGpu();
    void minDivergenceGpu();
    void normTMatrixGpu();
    void getWeightedCovGpu(alize::DoubleSquareMatrix &W, alize::DoubleVector &weight);
    void approximateTcTcGpu(alize::Matrix<double> &D, alize::Matrix<double> &Q);
    void normStatisticsGpu();
    void substractMplusTWGpu();
    void estimateWUbmWeightGpu(alize::DoubleSquareMatrix &W);
    void estimateWEigenDecompositionGpu(alize::Matrix<double> &D, alize::Matrix<double> &Q);
    void estimateAandCMultiGpu(gmmiv_comm *comm, int world, int rank, unsigned long nSessionsAllRanks);
};

class PldaTest {                                             // PldaTools.h:560-574
  public:
    unsigned long _vectSize, _n_models, _n_test_segments;
    alize::Matrix<double> _models, _segments, _scores;
    alize::BoolMatrix _trials;
    gmmiv_ctx *_gpu;
    void cosineDistanceGpu();
    void mahalanobisDistanceGpu(alize::DoubleSquareMatrix &Mah);
    void twoCovScoringGpu(alize::DoubleSquareMatrix &W, alize::DoubleSquareMatrix &B);
    void pldaScoringGpu(unsigned long rankF, alize::Matrix<double> &modelSums, std::vector<int64_t> &nsess, alize::Matrix<double> &FTJF);
    void applyTrialsGpu();
};

class PldaDev {                                              // PldaTools.h: class PldaDev
  public:
    unsigned long _vectSize, _n_speakers, _n_sessions;
    alize::Matrix<double> _data;
    alize::RealVector<double> _mean;
    alize::Matrix<double> _speaker_means;
    alize::ULongVector _session_per_speaker;
    gmmiv_ctx *_gpu;
    std::vector<int64_t> sps64() const;                      // _session_per_speaker as int64_t
    void computeAllGpu();
    void computeCovMatGpu(alize::DoubleSquareMatrix &Sigma, alize::DoubleSquareMatrix &W, alize::DoubleSquareMatrix &B);
    void computeWccnCholGpu(alize::DoubleSquareMatrix &WCCN);
    void computeMahalanobisGpu(alize::DoubleSquareMatrix &M);
    void computeScatterMatGpu(alize::DoubleSquareMatrix &SB, alize::DoubleSquareMatrix &SW);
    void computeLDAGpu(alize::Matrix<double> &ldaMat, long ldaRank, alize::DoubleSquareMatrix &W, alize::DoubleSquareMatrix &B);
    void sphericalNuisanceNormalizationIterationGpu(bool sphNorm, alize::DoubleSquareMatrix &sphNormMat);
};

class JFAAcc {                                               // AccumulateJFAStat.h: class JFAAcc
  public:
    unsigned long _vectSize, _n_distrib, _svSize, _rankEV, _rankEC, _n_speakers, _n_sessions;
    alize::RealVector<double> _ubm_means, _ubm_invvar;
    alize::Matrix<double> _matN, _N_h, _F_X, _F_X_h, _V, _matU, _Y, _matX, _Z, _Cev, _Cec;
    alize::DoubleVector _D;
    gmmiv_ctx *_gpu;
    std::vector<double> _vEvTPacked, _uEuTPacked, _aevPacked, _aecPacked;
    std::vector<int64_t> _speakerOfSession, _firstSessionOfSpeaker;   // from JFATranslate (sessions of a speaker are consecutive)
    void estimateVEVTGpu();
    void estimateYandVGpu();
    void estimateXandUGpu();
    void estimateYGpu();
    void updateVestimateGpu();
    void substractMplusDZGpu();
    void substractMplusVYGpu();
    void substractMplusVYplusDZGpu();
    void substractUXGpu();
    void estimateZMAPGpu(double tau);
    void estimateZandDGpu();
};

class TopGauss {                                             // TopGauss.h
  public:
    unsigned long _nt, _nbgcnt;
    alize::ULongVector _nbg, _idx;
    alize::DoubleVector _snsw, _snsl;
    double computeGpu(gmmiv_ctx *ctx, gmmiv_gmm *g, alize::FeatureServer &fs, alize::SegCluster &segs, double topD, int cap, double minLLK,
                      double maxLLK);
    double getGpu(gmmiv_ctx *ctx, gmmiv_gmm *g, alize::FeatureServer &fs, alize::SegCluster &segs, double minLLK, double maxLLK);
};
