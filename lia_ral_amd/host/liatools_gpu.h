// liatools_gpu.h -- C++ host side above the C ABI (include/gmmiv.h): the LIA_SpkTools driver
// functions of the hot path, same names / argument meaning / error behaviour as the reference,
// with the per-frame ALIZE calls replaced by batched calls into libgmmiv (HIP, gfx950).
//
// The ALIZE container types are not available (alize-core is external to LIA_RAL), so minimal
// stand-ins carry the same information: Seg/SegCluster (begin, length, source), FeatureBuffer
// (a FeatureServer with featureServerBufferSize = ALL_FEATURES, resident in HBM), MixtureGD
// (weights, means, covariances + computeAll()).  Errors throw liagpu::Exception, which a tool
// driver catches and prints exactly like `catch (Exception& e) { cout << e.toString(); }`
// (LIA_SpkDet/TrainWorld/src/TrainWorld.cpp:187-190).
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gmmiv.h"

namespace liagpu {

struct Exception : std::runtime_error {
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
    std::string toString() const { return std::string("[gmmiv::Exception] ") + what(); }
};

// SegTools.cpp:265-271: a label "begin_s end_s" selects frames begin..end INCLUSIVE (length = end-begin+1)
struct Seg {
    unsigned long begin = 0, length = 0;
    unsigned long source = 0; // index of the feature file in the FeatureBuffer
    unsigned long labelCode = 0; // Seg::labelCode(): the multi-cluster baggedSegments tags a segment with the component it was drawn for
};
typedef std::vector<Seg> SegCluster;
unsigned long totalFrame(const SegCluster &c);
unsigned long timeToFrameIdx(double time, double frameLength);   // SegTools.cpp:135-142
double frameIdxToTime(unsigned long idx, double frameLength);    // SegTools.cpp:143-148
Seg segFromLabel(double begin_s, double end_s, double frameLength, unsigned long source = 0);

class GpuServer; // context owner (the StatServer/MixtureServer pair of the reference)

// A Matrix<double> / DoubleVector of the reference that LIVES ON THE DEVICE.  The accumulator classes below (EMAcc, TVAcc,
// JFAAcc, PldaDev) keep their statistics, matrices and accumulators in these: the C ABI is called with device pointers, so
// nothing crosses PCIe between the steps of a training loop -- at BASELINE's T-matrix configuration one rank holds F (6.1 GB),
// TETt and A (1.3 GB each); staging them per call would cost more than the compute.  A host copy exists only while somebody
// looks at it: host() downloads when the device copy is newer and assumes the caller may write (the next dev() uploads it
// again), chost() is the read-only view.  All copies / memsets are enqueued on the context's stream, i.e. in order with the
// kernels of the C ABI calls; only transfers that involve host memory wait for the stream.
class DVec {
  public:
    DVec() = default;
    explicit DVec(GpuServer &srv) : _srv(&srv) {}
    ~DVec();
    DVec(const DVec &) = delete;
    DVec &operator=(const DVec &) = delete;
    void bind(GpuServer &srv) { _srv = &srv; }
    size_t size() const { return _n; }
    bool empty() const { return _n == 0; }
    void assign(size_t n, double v = 0.0);            // vector::assign: n elements equal to v (device-side for v == 0)
    void set(const double *h, size_t n);              // upload
    void set(const std::vector<double> &h) { set(h.data(), h.size()); }
    void copyFrom(const DVec &o);                     // device-to-device (storeAccs / restoreAccs)
    void swap(DVec &o);
    double *dev();                                    // read-write device view (uploads a modified host copy first)
    const double *cdev() const;                       // read-only device view
    std::vector<double> &host();                      // read-write host view (downloads if the device copy is newer)
    const std::vector<double> &chost() const;         // read-only host view
    void get(double *h, size_t n, size_t offset = 0) const; // download a range without keeping a host copy

  private:
    void sync() const;
    void *st() const;
    void reserve(size_t n);
    GpuServer *_srv = nullptr;
    mutable double *_d = nullptr;
    size_t _n = 0, _cap = 0;
    mutable std::vector<double> _h;
    mutable bool _hostValid = false;  // _h mirrors the device copy
    mutable bool _hostDirty = false;  // _h may have been written through host(): the device copy is stale
};

// FeatureServer(ALL_FEATURES): all frames of all sources, float32 like SPro files, device resident.
class FeatureBuffer {
  public:
    FeatureBuffer(GpuServer &srv, const float *frames, unsigned long nFrames, unsigned long vectSize,
                  const std::vector<unsigned long> &sourceFirstFrame = std::vector<unsigned long>(1, 0));
    ~FeatureBuffer();
    unsigned long getVectSize() const { return _d; }
    unsigned long getFeatureCount() const { return _n; }
    unsigned long getFirstFeatureIndexOfASource(unsigned long s) const { return _first.at(s); }
    const float *device() const { return _dev; }
    unsigned long unusableFrames() const { return _unusable; } // frames with a NaN / infinite / absurd value, counted once at upload
    // device matrix [n x D] of the frames selected by the cluster, in cluster order.  The selection is ENQUEUED on the server's
    // stream (run table from pinned memory + k_gather_runs) and the pointer is valid in stream order: every gmmiv call of the same
    // server that follows sees the frames; the host does not wait.  A cluster that is one contiguous run is returned in place.
    // The matrix is overwritten by the next select() -- also in stream order, after the kernels that still read it.
    const float *select(const SegCluster &c, unsigned long &nSelected);
    GpuServer &server() { return _srv; }

  private:
    unsigned long buildRuns(const SegCluster &c, unsigned long &nSelected, unsigned long &firstFrame, size_t &nPieces);
    GpuServer &_srv;
    float *_dev = nullptr, *_sel = nullptr;
    unsigned long _n, _d, _selCap = 0, _unusable = 0;
    std::vector<unsigned long> _first;
    int64_t *_hRuns = nullptr, *_dRuns = nullptr; // run table: pinned host copy, device copy
    size_t _runsCap = 0, _dRunsCap = 0;
    void *_runsCopied = nullptr;                  // hipEvent_t: the last upload of the pinned table has completed
};

// MixtureGD / DistribGD: weight(c), getMean/getCov/getCovInv, setMean/setCov, computeAll()
class MixtureGD {
  public:
    MixtureGD(unsigned long distribCount, unsigned long vectSize);
    unsigned long getDistribCount() const { return _c; }
    unsigned long getVectSize() const { return _d; }
    double &weight(unsigned long c) { return _w[c]; }
    double weight(unsigned long c) const { return _w[c]; }
    double getMean(unsigned long c, unsigned long i) const { return _mean[c * _d + i]; }
    double getCov(unsigned long c, unsigned long i) const { return _cov[c * _d + i]; }
    double getCovInv(unsigned long c, unsigned long i) const { return _covInv[c * _d + i]; }
    void setMean(unsigned long c, double v, unsigned long i) { _mean[c * _d + i] = v; }
    void setCov(unsigned long c, double v, unsigned long i) { _cov[c * _d + i] = v; }
    // a model file stores covInv: keep the file's bits (1 / (1 / v) need not round back to v)
    void setCovInv(unsigned long c, double v, unsigned long i) { _covInv[c * _d + i] = v; _cov[c * _d + i] = 1.0 / v; }
    void computeAll(); // covInv = 1/cov (cst, det are derived on the device)
    std::vector<double> &weights() { return _w; }
    std::vector<double> &means() { return _mean; }
    std::vector<double> &covs() { return _cov; }
    const std::vector<double> &covInvs() const { return _covInv; }

  private:
    unsigned long _c, _d;
    std::vector<double> _w, _mean, _cov, _covInv;
};

// Owns the gmmiv context (one per GPU).  createAndStoreMixtureStat() -> EMAcc / LLKAcc below.
class GpuServer {
  public:
    explicit GpuServer(int device = 0);
    ~GpuServer();
    gmmiv_ctx *ctx() { return _ctx; }
    void *stream() { return gmmiv_ctx_stream(_ctx); } // hipStream_t of the context: the host layer's own copies are ordered on it
    void sync() { check(gmmiv_ctx_sync(_ctx)); }
    // grow-only device workspace of the host layer itself (slot 0..7): the per-file buffers of computeTestLLR live here instead of
    // a hipMalloc / hipFree pair -- two device synchronisations -- per test file.  Contents do not survive a growth.
    void *workspace(int slot, size_t bytes);
    void check(int rc) const; // throws Exception(gmmiv_last_error()) on rc != 0

  private:
    gmmiv_ctx *_ctx = nullptr;
    void *_ws[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t _wsBytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Scope of ONE frame-consuming call of the C ABI on `srv`'s context with frames of `fs`: the per-call screening pass is skipped
// (assume_finite) iff that buffer was found free of unusable frames at upload, or the user had set the option; the previous
// value comes back at scope exit.  The decision follows the buffer actually read, which may belong to another server.
class FiniteScope {
  public:
    FiniteScope(GpuServer &srv, const FeatureBuffer &fs);
    ~FiniteScope();
    FiniteScope(const FiniteScope &) = delete;
    FiniteScope &operator=(const FiniteScope &) = delete;

  private:
    gmmiv_ctx *_ctx;
    long _prev;
};

// Device copy of a MixtureGD
class DeviceMixture {
  public:
    DeviceMixture(GpuServer &srv, const MixtureGD &m);
    ~DeviceMixture();
    void update(const MixtureGD &m);
    gmmiv_gmm *handle() const { return _g; }
    GpuServer &server() { return _srv; }
    unsigned long getDistribCount() const { return _c; }
    unsigned long getVectSize() const { return _d; }

  private:
    GpuServer &_srv;
    gmmiv_gmm *_g = nullptr;
    unsigned long _c = 0, _d = 0;
};

// MixtureStat in EM mode: resetEM / computeAndAccumulateEM (batched) / getEM / getEMFeatureCount / addAccEM
class EMAcc {
  public:
    EMAcc(DeviceMixture &dm, const MixtureGD &model);
    void resetEM();
    double getEMFeatureCount() const;   // sum of the frame weights (2 doubles come back from the device)
    double getAccumulatedLLK() const;
    MixtureGD getEM() const; // ML weights / means / covariances; occ == 0 keeps the model's values
    void setModel(const MixtureGD &model) { _model = model; } // the model the next getEM() falls back to (createAndStoreMixtureStat(*world) of the next iteration)
    void addAccEM(const EMAcc &o);
    std::vector<double> &flat() { return _acc.host(); } // host view of the flat accumulator (downloaded on demand)
    DVec &acc() { return _acc; }                         // the device-resident accumulator = the all-reduce payload
    DeviceMixture &mixture() { return _dm; }

  private:
    DeviceMixture &_dm;
    MixtureGD _model;
    DVec _acc;
};

// FrameAccGD
struct FrameAccGD {
    std::vector<double> acc; // [sum | sumsq | n]
    unsigned long vectSize = 0;
    unsigned long getCount() const { return acc.empty() ? 0 : (unsigned long)acc.back(); }
    std::vector<double> getMeanVect() const;
    std::vector<double> getCovVect() const; // biased: sumsq/n - mean^2
};

// ---- AccumulateStat.h --------------------------------------------------------------------------
// accumulateStatEM (AccumulateStat.cpp:103-152): returns sum_t log lk_t over the cluster
double accumulateStatEM(FeatureBuffer &fs, EMAcc &emAcc, const SegCluster &selectedSegments);
double accumulateStatEM(FeatureBuffer &fs, EMAcc &emAcc, const SegCluster &selectedSegments, double weight);
// accumulateStatLLK (AccumulateStat.cpp:69-94): mean clamped llk over the cluster (getMeanLLK)
double accumulateStatLLK(FeatureBuffer &fs, DeviceMixture &m, const SegCluster &selectedSegments, double minLLK,
                         double maxLLK);
// MixtureStat in LLK mode: resetLLK / computeAndAccumulateLLK(f, weight) (batched) / getMeanLLK = sum w llk / sum w -- the accumulator the
// meanLikelihood family drives over one cluster, over several streams (GeneralTools.cpp:589-607) and over weighted feature servers
// (`decision[nbFs]` as the frame weight: AccumulateStat.cpp:344-379, GeneralTools.cpp:610-624)
struct LLKAcc {
    double sumLLK = 0.0, sumWeight = 0.0;
    void resetLLK() { sumLLK = sumWeight = 0.0; }
    double getMeanLLK() const { return sumWeight > 0.0 ? sumLLK / sumWeight : 0.0; }
    double getAccumulatedLLK() const { return sumLLK; }
    double getAccumulatedLLKFeatureCount() const { return sumWeight; }
};
void accumulateStatLLK(LLKAcc &llkAcc, FeatureBuffer &fs, DeviceMixture &m, const SegCluster &selectedSegments, double weight, double minLLK,
                       double maxLLK);
struct TrainStream;
double meanLikelihood(const std::vector<TrainStream> &streams, DeviceMixture &model, double minLLK, double maxLLK);                  // :599-607
double meanLikelihood(const std::vector<TrainStream> &streams, DeviceMixture &model, const std::vector<double> &decision, double minLLK,
                      double maxLLK);                                                                                                  // :610-624
// accumulateStatFrame (AccumulateStat.cpp:387-410)
void accumulateStatFrame(FrameAccGD &frameAcc, FeatureBuffer &fs, const SegCluster &selectedSegments);

// ---- TrainTools.h ------------------------------------------------------------------------------
struct TrainCfg { // TrainTools.cpp:67-93
    double initVarianceFlooring = 0.0, initVarianceCeiling = 10.0, finalVarianceFlooring = 0.0, finalVarianceCeiling = 10.0;
    unsigned long nbTrainIt = 1;
    double baggedFrameProbability = 1.0;
    unsigned long baggedMinimalLength = 3, baggedMaximalLength = 7, initRand = 0;
    // componentReduction / targetMixtureDistribCount (:86-91): after every iteration keep the nbTop heaviest components, nbTop going
    // linearly from the initial count to the target (TrainTools.cpp:1078-1097)
    bool componentReduction = false;
    unsigned long targetDistribCount = 0;
    // normalizeModel (:76-84): after every iteration move the mixture to global mean 0 / variance 1 (normalizeMixture, :287-315);
    // normalizeModelNbIt only applies with normalizeModelMeanOnly
    bool normalizeModel = false, normalizeModelMeanOnly = false;
    unsigned long normalizeModelNbIt = 1;
    // not a reference parameter: when set, receives the wall time (ms) of every iteration, measured from the end of the previous
    // iteration's model update to the end of this one's (what bench.py's host_layer block reports)
    std::vector<double> *iterationMs = nullptr;
};
double setItParameter(double begin, double end, int nbIt, int it);                                 // :560-564
void varianceControl(MixtureGD &model, double flooring, double ceiling, const std::vector<double> &covSignal); // :567-587
unsigned long computeMeanCov(FeatureBuffer &fs, const SegCluster &seg, std::vector<double> &mean, std::vector<double> &cov); // :593-611
// GeneralTools.cpp:455-510 with glibc rand() (baggedFrame :309-314); caller seeds with srand()
void baggedSegments(const SegCluster &selectedSegments, SegCluster &baggedFrameSegment, double baggedProbability,
                    unsigned long minimumLength, unsigned long maximumLength);
// The multi-selection form (GeneralTools.cpp:330-390): ONE walk over the input segments, nbBagged independent draws per chunk;
// a chunk drawn for component idx is appended with labelCode = idx (rand() order: chunk-major, component-minor)
void baggedSegments(const SegCluster &selectedSegments, SegCluster &baggedSeg, unsigned long nbBagged, double baggedProbability,
                    unsigned long minimumLength, unsigned long maximumLength);
// mixtureInit: the start-from-scratch model of TrainWorld (random picking of frames per component; means = the picked
// frames' means, covariances = globalCov, equal weights).  glibc srand / rand in the reference's order, like baggedSegments.
//   multi-stream form, TrainTools.cpp:674-766 -- the one TrainWorld.cpp:177 calls -- for ONE stream of weight streamWeight:
//     p = nbFrameToSelect * weight / totalFrame (folded into several bagging passes when > 1, :700-708 -- as written that fold
//     only terminates for p >= ~4.9; for 1 < p < 4.9 the reference spins forever and this layer throws instead), seed
//     srand((stream + 1) * 100 + baggedIt + 1), one multi-selection baggedSegments pass per iteration;
//   single-stream form, TrainTools.cpp:619-672: p = baggedFrameProbabilityInit / distribCount, seed
//     srand((indg + 1) * (baggedIt + 1)) and one plain baggedSegments pass PER COMPONENT.
// A component that drew no frame has no mean: Exception (FrameAccGD::getMeanVect on an empty accumulator is undefined in
// the reference).  frameCount (optional) receives the frames picked per component.
struct MixtureInitCfg { unsigned long baggedMinimalLength = 3, baggedMaximalLength = 7; double nbFrameToSelect = 50; double baggedFrameProbabilityInit = 0.0; };
struct TrainStream;
// all input streams (fsTab / segTab / weightTab), the form TrainWorld.cpp:177 calls; the one-stream overload below forwards here
void mixtureInit(const std::vector<TrainStream> &streams, MixtureGD &world, const std::vector<double> &globalCov, const MixtureInitCfg &cfg,
                 std::vector<unsigned long> *frameCount = nullptr);
void mixtureInit(FeatureBuffer &fs, const SegCluster &selectedSegments, double streamWeight, MixtureGD &world,
                 const std::vector<double> &globalCov, const MixtureInitCfg &cfg, std::vector<unsigned long> *frameCount = nullptr);
void mixtureInitSingleStream(FeatureBuffer &fs, MixtureGD &world, const SegCluster &selectedSegments, const std::vector<double> &globalCov,
                             const MixtureInitCfg &cfg, std::vector<unsigned long> *frameCount = nullptr);
// trainModelStream (TrainTools.cpp:1030-1110), single stream; returns the per-iteration mean llk
// ("llkPreviousIt").  allReduce, when given, sums the flat accumulator over ranks (RCCL/xGMI).
// Multi-GPU: frames sharded per rank, ONE all-reduce of the flat EM accumulator per iteration, the collective twin of
// MixtureStat::addAccEM (AccumulateStat.cpp:286-292).  The default is the C ABI's own RCCL communicator (comm, on the device
// accumulator); AllReduceFn is the hook for another transport (it gets the HOST view of the accumulator).
typedef void (*AllReduceFn)(double *buf, size_t n, void *user);
// one input stream of TrainWorld ("inputStreamList" / "weightStreamList", TrainWorld.cpp:123-137): fsTab[i], segTab[i], weightTab[i].
// Every stream's FeatureBuffer lives on the same GpuServer.  weightTab defaults to 1 / nbStream (reserveMem, TrainWorld.cpp:85).
struct TrainStream { FeatureBuffer *fs = nullptr; const SegCluster *segs = nullptr; double weight = 1.0; };
// The stream form (TrainTools.cpp:1030-1110): per iteration and stream, baggedProba = p * nbTotalFrame * weight / totalFrame(stream)
// (folded into nbBaggedIt passes when > 1), seed ((trainIt+1+initRand)*200) + ((stream+1)*20) + (baggedIt+1), ONE accumulator over
// all streams; then getEM, varianceControl, componentReduction, normalizeModel.  `world` may come back with fewer components.
// The host prepares the selection of the next pass while the kernels of the current one run (rand() order unchanged: every pass
// seeds the generator itself).
std::vector<double> trainModelStream(const TrainCfg &cfg, const std::vector<TrainStream> &streams, const std::vector<double> &globalCov,
                                     MixtureGD &world, AllReduceFn allReduce = nullptr, void *user = nullptr, gmmiv_comm *comm = nullptr);
// component selection and model normalisation used by it (TrainTools.cpp:186-227, :240-315; TabWeight: GeneralTools.h:145-183)
std::vector<unsigned long> sortByWeight(const MixtureGD &model);     // TabWeight::_sortByWeight: qsort, heaviest first
unsigned long selectComponent(std::vector<bool> &selectCompA, unsigned long nbTop, const MixtureGD &inputM); // the nbTop heaviest
unsigned long selectComponent(std::vector<bool> &selectCompA, double wFactor, const MixtureGD &inputM);      // weight >= wFactor
double reduceModel(const std::vector<bool> &selectCompA, const MixtureGD &inputM, MixtureGD &outputM);        // returns the kept weight
void normalizeWeights(MixtureGD &outputM);
void mixtureFusion(const MixtureGD &mixt, std::vector<double> &mean, std::vector<double> &cov, double &wres);
void normalizeMixture(MixtureGD &mixt, const std::vector<double> &meanSignal, const std::vector<double> &covSignal, bool zeroOne,
                      unsigned long nbIt, bool meanOnly);
// global mean / covariance over several streams (computeMeanCov(config, fsTab, segTab, nbStream, ...), TrainTools.cpp:593-601)
unsigned long computeMeanCov(const std::vector<TrainStream> &streams, std::vector<double> &mean, std::vector<double> &cov);
std::vector<double> trainModelStream(const TrainCfg &cfg, FeatureBuffer &fs, const SegCluster &selectedSegments,
                                     const std::vector<double> &globalCov, MixtureGD &world, gmmiv_comm *comm);
std::vector<double> trainModelStream(const TrainCfg &cfg, FeatureBuffer &fs, const SegCluster &selectedSegments,
                                     const std::vector<double> &globalCov, MixtureGD &world,
                                     AllReduceFn allReduce = nullptr, void *user = nullptr, gmmiv_comm *comm = nullptr);

// ---- TrainTarget: MAP adaptation (TrainTools.cpp:445-489 computeMAPOccDep, :871-904 adaptModel) ----
struct MAPCfg { // MAPCfg::MAPCfg, TrainTools.cpp:95-147
    std::string method = "MAPOccDep"; // MAPAlgo: MAPOccDep, MAPModelBased (regulation factors), MAPConst, MAPConst2 (constant alpha)
    unsigned long nbTrainIt = 1;
    double baggedFrameProbability = 1.0;
    bool meanAdapt = true, varAdapt = false, weightAdapt = false;
    double meanReg = 16.0, varReg = 16.0, weightReg = 16.0; // MAPRegFactorMean / Var / Weight
    double meanAlpha = 0.75;                                  // MAPAlphaMean: a-priori probability of the init model (MAPConst / MAPConst2)
    bool normalizeModel = false, normalizeModelMeanOnly = false; // :125-135, applied after every iteration's MAP step (:898)
    unsigned long normalizeModelNbIt = 1;
};
void computeMAPOccDep(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg, double frameCount);              // :445-489
void computeModelBasedMAPOccDep(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg, double frameCount);    // :491-536 (the same arithmetic)
void computeMAPConst(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg);                                  // :356-384 (means only)
void computeMAPConst2(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg);                                 // :390-420 (means only, weight-balanced)
// computeMAP (:543-556): dispatch on cfg.method; an unknown method leaves the client as it is (the reference prints a warning)
void computeMAP(const MixtureGD &initModel, MixtureGD &client, unsigned long frameCount, const MAPCfg &cfg);
// client model = MAP(aprioriModel, EM estimate on the selected frames), nbTrainIt times
void adaptModel(FeatureBuffer &fs, const SegCluster &selectedSegments, const MixtureGD &aprioriModel,
                MixtureGD &clientMixture, const MAPCfg &mapCfg);

// ---- ComputeTest (LIA_SpkDet/ComputeTest/src/ComputeTest.cpp:129-215) -----------------------------
// LLR of each client against the world for one test file: per segment when segmentalMode, else one
// per file.  out[(seg or 0) * nClients + i] = mean llk_client - mean llk_world.
std::vector<double> computeTestLLR(FeatureBuffer &fs, const SegCluster &selectedSegments, DeviceMixture &world,
                                   std::vector<DeviceMixture *> &clients, int topDistribsCount, bool complete,
                                   double minLLK, double maxLLK, bool segmentalMode);
// WindowLLR (LIA_SpkTools/src/UnsupervisedTools.cpp:70-145): sliding window over the per-frame LLRs of ComputeTest
// (ComputeTest.cpp:165-178).  One WindowOut per position at which the window is full.
struct WindowLLR {
    WindowLLR(unsigned long size, unsigned long dec, unsigned long nClient);
    void dec(unsigned long idxFrame);                        // :119-135
    void accLLR(unsigned long clientIdx, double llr);        // :136-139
    double getLLR(unsigned long clientIdx) const { return _acc[clientIdx] / (double)_size; }
    bool isEnd() const { return _count == _size; }
    unsigned long getIdxBegin() const { return _idx[_bIdx]; }
    unsigned long getIdxEnd() const { return _idx[(_bIdx + _count - 1) % _size]; }
    unsigned long _size, _dec, _nClient, _bIdx, _count;
    std::vector<unsigned long> _idx;
    std::vector<double> _acc, _llr; // _llr [size x nClient]
};
struct WindowOut { unsigned long idxBegin, idxEnd; std::vector<double> llr; /* one per client */ };
// The full frame loop of ComputeTest.cpp:154-207: worldDecime (DETERMINE_TOP_DISTRIBS on every worldDecime-th frame
// of a segment, USE_TOP_DISTRIBS with the last top set on the others) and the optional windowed LLRs
// (windowSize == 0: off).  Returns the same file / segment LLRs as computeTestLLR.
std::vector<double> computeTestLLR(FeatureBuffer &fs, const SegCluster &selectedSegments, DeviceMixture &world,
                                   std::vector<DeviceMixture *> &clients, int topDistribsCount, bool complete,
                                   double minLLK, double maxLLK, bool segmentalMode, unsigned long worldDecime,
                                   unsigned long windowSize, unsigned long windowDec, std::vector<WindowOut> *windows);

// ---- TopGauss.h (LIA_SpkTools/src/TopGauss.cpp) -----------------------------------------------------
// The per-frame Gaussian selection the factor-analysis tools compute once per feature file and cache on disk.
class TopGauss {
  public:
    // compute (:136-198): DETERMINE_TOP_DISTRIBS on every selected frame with a list of topDistribsCount entries (any count up to the model's), then
    // topGauss >= 1: the (unsigned long)topGauss heaviest Gaussians of every frame; topGauss < 1: Gaussians until their cumulative
    // likelihood passes topGauss * exp(llk) -- a variable count per frame.  Returns getMeanLLK() of the DETERMINE pass.
    double compute(DeviceMixture &ubm, FeatureBuffer &fs, const SegCluster &selectedSegments, double topGauss, int topDistribsCount,
                   bool complete, double minLLK, double maxLLK);
    // get (:275-316): mean llk of `ubm` (any model with the UBM's component order) on the stored selection (USE_TOP_DISTRIBS with
    // setTopDistribIndexVector(index, snsw, snsl) per frame)
    double get(DeviceMixture &ubm, FeatureBuffer &fs, const SegCluster &selectedSegments, bool complete, double minLLK, double maxLLK) const;
    // write / read (:200-224 / :76-98), the reference's binary layout on LP64: _nt, _nbgcnt (unsigned long, 8 bytes, native order),
    // _nbg[_nt] (unsigned long), _idx[_nbgcnt] (unsigned long), _snsw[_nt], _snsl[_nt] (double)
    void write(const std::string &path) const;
    void read(const std::string &path);
    unsigned long frameToIdx(unsigned long f) const; // :68-74
    unsigned long nt() const { return _nt; }
    unsigned long nbgcnt() const { return _nbgcnt; }
    unsigned long nbCapped() const { return _capped; } // frames whose likelihood mass was not reached within topDistribsCount entries
    const std::vector<unsigned long> &nbg() const { return _nbg; }
    const std::vector<unsigned long> &idx() const { return _idx; }
    const std::vector<double> &snsw() const { return _snsw; }
    const std::vector<double> &snsl() const { return _snsl; }

  private:
    unsigned long _nt = 0, _nbgcnt = 0, _capped = 0;
    std::vector<unsigned long> _nbg, _idx;
    std::vector<double> _snsw, _snsl;
};

// ---- GmmTokenizer (LIA_Utils/GmmTokenizer/src/GmmTokenizer.cpp): the two consumers of getTopDistribIndexVector() that ship golden
// outputs (test/test1.sym.ref, test/mce_matrix.mat.ref -- KAT-5 of tests/golden) ------------------------------------------------
// computeSymbols (:99-126): DETERMINE_TOP_DISTRIBS with a list of topDistribsCount entries on every selected frame, v[0].idx appended
// to `stream` -- one symbol per frame, in cluster order.
void computeSymbols(const SegCluster &selectedSegments, FeatureBuffer &fs, DeviceMixture &world, std::vector<unsigned long> &stream,
                    int topDistribsCount = 1, double minLLK = -200.0, double maxLLK = 200.0);
// computeConfusionMatrix (:69-97): mce_matrix(v[0].idx, v[i].idx)++ for i < nBest on every selected frame; mce_matrix is
// [distribCount x distribCount], row-major, ACCUMULATED into (the caller sizes and zeroes it, like mce_matrix.setDimensions at :144).
// nBest = topDistribsCount there (:132), so the list length and the loop bound are one number.
void computeConfusionMatrix(const SegCluster &selectedSegments, FeatureBuffer &fs, DeviceMixture &world, unsigned long nBest,
                            std::vector<unsigned long> &mce_matrix, double minLLK = -200.0, double maxLLK = 200.0);

// ---- EnergyDetector (LIA_SpkDet/EnergyDetector/src/EnergyDetector.cpp): a two / three-Gaussian model of the energy coefficient trained
// by FULL EM (weights, means AND variances) from a FIXED init -- no rand() -- then a threshold on that coefficient.  The one reference tool
// whose output (test/test1.validate.enr.lbl, KAT-6 of tests/golden, "assumed") depends on the variance estimate of getEM.
struct EnergyDetectorCfg {
    unsigned long nbTrainIt = 10, mixtureDistribCount = 3;
    double varianceFlooring = 0.5, varianceCeiling = 10.0, alpha = 0.0;
    std::string thresholdMode = "meanStd"; // :199-203; "weight" needs alize-core's Histo (not in the LIA_RAL tree): Exception
};
void energyMixtureInit(MixtureGD &world);                    // :163-183: means from -2 in steps of 4 / (distribCount - 1), covariances 1, equal weights
unsigned long findMaxEnergyDistrib(const MixtureGD &mixt);   // :83-93
// selectFrames (:118-157): segments of consecutive frames of `selectedSeg` whose coefficient 0 exceeds the threshold; begin / length count
// frames of the SELECTION (the reference's `ind`), and a run that reaches the end of an input segment is one frame longer than a run that
// ends inside it (:144 vs :151 -- reproduced as written).  energy[t] = coefficient 0 of frame t of the file.  Returns the frames above.
unsigned long selectFrames(const std::vector<float> &energy, double threshold, const SegCluster &selectedSeg, SegCluster &outputSeg);
// energyDetector (:190-285): globalMeanCov -> energyMixtureInit -> nbTrainIt x (accumulateStatEM, getEM, varianceControl) -> threshold
// = mean - alpha * sqrt(cov) of the highest-mean component -> selectFrames.  energyModel / threshold (optional) receive the trained model
// and the threshold.
SegCluster energyDetector(FeatureBuffer &fs, const SegCluster &selectedSegments, const EnergyDetectorCfg &cfg, MixtureGD *energyModel = nullptr,
                          double *threshold = nullptr);

// ---- AccumulateTVStat.h ----------------------------------------------------------------------------
class TVAcc {
  public:
    // ndx lines = statistics rows; row u owns the frames of segs[u] (TVAcc::_init, :129-196)
    TVAcc(GpuServer &srv, const MixtureGD &ubm, unsigned long rankT, unsigned long nSpeakers);
    void computeAndAccumulateTVStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerLine); // :281-351
    // the same with the reference's file -> line map (TVTranslate::locIndices, :318-346): segsPerFile[f] = selected segments of
    // feature file f, filesOfLine[l] = the files listed on ndx line l (a file may be listed on several lines: it is evaluated
    // once and its statistics are added to each)
    void computeAndAccumulateTVStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerFile,
                                    const std::vector<std::vector<unsigned long> > &filesOfLine);
    // initT, randomInitLaw "normal" (:729-748): T(i,j) = boxMuller(0,1) * (sum_k invvar_k) * 0.001 with the Box-Muller chain of
    // ScoreWarp.cpp:68-81 on glibc rand() (no srand: the caller's seed state, 1 by default); NaN / Inf draws are redrawn.  The
    // "uniform" law calls Matrix::randomInit of alize-core (not in the LIA_RAL tree): not reproduced, throws.
    void initT(const std::string &randomInitLaw = "normal");
    void substractM();                 // :1088-1105
    void estimateTETt();               // :777-805
    void estimateW();                  // :2114-2169
    void estimateAandC();              // :1702-1795
    void updateTestimate();            // :974-1005
    void minDivergence();              // :2056-2099
    void orthonormalizeT();            // :1548-1596
    // approximate extractors (IvExtractor modes ubmWeight / eigenDecomposition, IvExtractor.cpp:150-360)
    void normStatistics();             // :1225-1242
    void substractMplusTW();           // :1379-1399
    void normTMatrix();                // :1600-1609
    void getWeightedCov(std::vector<double> &W, const std::vector<double> &weight);       // :2837-2855, W [R x R]
    void approximateTcTc(std::vector<double> &D, const std::vector<double> &Q);            // :3116-3136, D [C x R] accumulated
    void estimateWUbmWeight(const std::vector<double> &W);                                 // :2348-2396 (zeroes _W first)
    void estimateWEigenDecomposition(const std::vector<double> &D, const std::vector<double> &Q); // :2566-2609 (accumulates into _W)
    void resetTmpAcc();                // :620-629
    void loadT(const std::vector<double> &T);
    void setStats(const std::vector<double> &N, const std::vector<double> &F);   // loadN / loadF (upload, no host copy kept)
    void setStats(const double *N, const double *F);
    // TotalVariability reloads N and F from disk at every iteration because substractM works in place
    // (TotalVariability.cpp:152-153); here the pristine statistics are kept on the device instead
    void storeStats();
    void restoreStats();
    void restoreStatsAndSubstractM();  // both steps of an iteration's start in one pass over F (gmmiv_tv_subtract_m_to)
    // Multi-GPU form of updateTestimate for utterance-sharded statistics: reduce-scatter of A / Cmx by blocks
    // of Gaussians, T_c = A_c^-1 Cmx_c on the rank's own Gaussians, all-gather of T; R, r and meanW (sums) are all-reduced
    // for minDivergence, whose session count becomes the global one.  comm == NULL or one rank: plain updateTestimate.
    void updateTestimate(gmmiv_comm *comm, unsigned long nSessionsAllRanks);
    // Overlapped exchange (option; bitwise the results of the serial order): with a communicator set here, estimateAandC()
    // accumulates A in the reduce-scatter's (block-padded) send buffer and BEGINS that reduce-scatter from the "tv_a_ready" hook of
    // gmmiv_tv_estimate_a_and_c -- it runs on the communicator's side stream under the Cmx GEMM -- and updateTestimate(comm, n)
    // leaves the all-gather of T in flight for minDivergence(), which joins it from the "md_factored" hook (after R is factored).
    // A caller that skips minDivergence() calls finishT() before anything reads T.  comm == NULL: serial order again.
    void setOverlap(gmmiv_comm *comm);
    void finishT();
    // The EM sanity check of TotalVariability (TotalVariability.cpp:132 `if (_checkLLK) tvAcc.verifyEMLK(config)`):
    //   getMplusTW (:964-971)      Sp = ubm_means + T^T w_spk  -- for a LIST of rows at once: one device pass, Sp [rows x svSize]
    //   getSpeakerModel (:1533-1545) the UBM with those means (svToModel, SuperVectors.cpp:79-85: means only)
    //   getLLK (:1626-1651)        mean clamped log-likelihood of the selected frames under a model (computeAndAccumulateLLK loop)
    //   verifyEMLK (:1654-1688)    sum over the first maxLLKcomputed files of getLLK(file's segments, speaker model of its row);
    //                              rowOfFile[f] = the statistics row (ndx line) file f belongs to; perFile (optional) = each llk
    void getMplusTW(std::vector<double> &Sp, const std::vector<unsigned long> &rows);
    void getSpeakerModel(MixtureGD &mixture, unsigned long spk);
    double getLLK(const SegCluster &selectedSegments, const MixtureGD &model, FeatureBuffer &fs, double minLLK = -200.0, double maxLLK = 200.0);
    double verifyEMLK(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerFile, const std::vector<unsigned long> &rowOfFile,
                      unsigned long maxLLKcomputed, double minLLK = -200.0, double maxLLK = 200.0, std::vector<double> *perFile = nullptr);
    // all host views below download on demand (and re-upload before the next device step, they may be written through)
    std::vector<double> &getT() { return _T.host(); }
    std::vector<double> &getW() { return _W.host(); }
    std::vector<double> &getN() { return _statN.host(); }
    std::vector<double> &getF() { return _statF.host(); }
    std::vector<double> &getUbmMeans() { return _ubm_means.host(); }
    DVec &deviceT() { return _T; }
    DVec &deviceW() { return _W; }
    unsigned long getRankT() const { return _rankT; }

  private:
    GpuServer &_srv;
    MixtureGD _ubm;
    DeviceMixture _dubm;
    unsigned long _rankT, _n_speakers, _n_distrib, _vectSize, _svSize, _n_sessions_global;
    DVec _ubm_means, _ubm_invvar, _statN, _statF, _cN, _cF, _T, _W, _TETt, _A, _Cmx, _R, _r, _meanW;
    // overlapped exchange
    gmmiv_comm *_ovComm = nullptr;
    DVec _aMine, _tAll, _tMine;
    bool _aBegun = false, _tPending = false;
    static void hookAReady(void *self);
    static void hookMdFactored(void *self);
    void gatherIntoT();
};

// ---- AccumulateJFAStat.h: JFAAcc, M_{s,h} = m + V y_s + U x_h + D z_s (AccumulateJFAStat.cpp) ---------------------------
// Statistics rows: _matN / _F_X per speaker (ndx line), _N_h / _F_X_h per session (ndx element); sessions are grouped by
// speaker in ndx order.  L matrices are never stored: estimateAndInverseL_E{V,C} only mark the step, the build + inverse
// happens inside the estimate* call that consumes it, on the device (one workgroup per system).
class JFAAcc {
  public:
    JFAAcc(GpuServer &srv, const MixtureGD &ubm, unsigned long rankEV, unsigned long rankEC,
           const std::vector<unsigned long> &sessionsPerSpeaker);                          // _init, :168-300
    void computeAndAccumulateJFAStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerSession); // :515-577
    void setStats(const std::vector<double> &N, const std::vector<double> &N_h, const std::vector<double> &F_X,
                  const std::vector<double> &F_X_h);                                       // loadN / loadN_h / loadF_X / loadF_X_h, :1025-1068
    void storeAccs();                  // :3777-3784
    void restoreAccs();                // :3786-3793
    void resetTmpAcc();                // :863-897
    void loadEV(const std::vector<double> &V);      // :949-968
    void loadEC(const std::vector<double> &U);      // :983-995
    void loadD(const std::vector<double> &D);       // :1016-1023
    void initD(double regulationFactor);            // initDType "MAP", :1214-1218
    void estimateVEVT();               // :1266-1352
    void estimateUEUT();               // :1425-1508
    void estimateAndInverseL_EV() {}   // :1970-1996 (fused into estimateYandV / estimateY)
    void estimateAndInverseL_EC() {}   // :2137-2163 (fused into estimateXandU / estimateX)
    void estimateYandV();              // :2467-2511
    void estimateY();                  // :2867-2957
    void estimateXandU();              // :3040-3083
    void estimateX();                  // :3262-3351
    void estimateZandD();              // :3480-3516
    void estimateZ();                  // :3550-3573
    void estimateZMAP(double tau);     // :3576-3594
    void updateVestimate();            // :3597-3619
    void updateUestimate();            // :3622-3644
    void substractMplusDZ();           // :3805-3822   _F_X   -= N   (m + D z)
    void substractMplusVY();           // :3988-4005   _F_X   -= N   (m + V y)
    void substractUX();                // :4152-4172   _F_X   -= sum_h N_h (U x_h)
    void substractMplusVYplusDZ();     // :4400-4422   _F_X_h -= N_h (m + V y + D z) of the session's speaker
    void substractMplusUX();           // :4336-4364   _F_X   -= sum_h N_h (m + U x_h)
    void substractMplusDZByChannel();  // :3948-3976   _F_X_h -= N_h (m + D z) of the session's speaker
    void orthonormalizeV();            // :4700-4777
    void getMplusVYplusDZ(std::vector<double> &Sp, unsigned long spk); // :1926-1935
    std::vector<double> &getV() { return _V.host(); }
    std::vector<double> &getU() { return _matU.host(); }
    std::vector<double> &getD() { return _D.host(); }
    std::vector<double> &getY() { return _Y.host(); }
    std::vector<double> &getX() { return _matX.host(); }
    std::vector<double> &getZ() { return _Z.host(); }
    std::vector<double> &getN() { return _matN.host(); }
    std::vector<double> &getN_h() { return _N_h.host(); }
    std::vector<double> &getF_X() { return _F_X.host(); }
    std::vector<double> &getF_X_h() { return _F_X_h.host(); }
    unsigned long getNSpeakers() const { return _n_speakers; }
    unsigned long getNSessions() const { return _n_sessions; }

  private:
    GpuServer &_srv;
    MixtureGD _ubm;
    DeviceMixture _dubm;
    unsigned long _rankEV, _rankEC, _n_speakers, _n_sessions, _n_distrib, _vectSize, _svSize;
    std::vector<int64_t> _sess_begin, _owner;
    DVec _ubm_means, _ubm_invvar, _matN, _N_h, _F_X, _F_X_h, _cN, _cN_h, _cF_X, _cF_X_h;
    DVec _V, _matU, _D, _Y, _matX, _Z, _vEvT, _uEuT, _Aev, _Cev, _Aec, _Cec, _mdR, _mdr, _mdmw; // _md*: minimum-divergence sums, unused by JFA
};
// the three training tools around it (statistics and initial matrices already in the accumulator)
void eigenVoice(JFAAcc &jfaAcc, unsigned long nbIt, bool orthonormalizeV);   // EigenVoice.cpp:114-147
void eigenChannel(JFAAcc &jfaAcc, unsigned long nbIt);                        // EigenChannel.cpp:118-160
void estimateDMatrix(JFAAcc &jfaAcc, unsigned long nbIt);                     // EstimateDMatrix.cpp:143-206
// ComputeTest in the JFA framework, dot-product scoring (ComputeTest.cpp:303-358): every statistics row of jfaAcc is one test
// segment (one session, y = z = 0); returns scores[nTest x nClients] = <client supervector, channel-compensated mean statistics>
std::vector<double> computeTestDotProduct(GpuServer &srv, JFAAcc &jfaAcc, const std::vector<double> &clientSV, unsigned long nClients);

// ---- PldaTools.h: PldaDev, the development set of the i-vector back-end (PldaTools.cpp:274-2005) -------------
// _data [vectSize x n_sessions] (one i-vector per column), sessions grouped by speaker.
class PldaDev {
  public:
    PldaDev(GpuServer &srv, unsigned long vectSize, const std::vector<double> &data, const std::vector<unsigned long> &sessionPerSpeaker);
    unsigned long getVectSize() const { return _vectSize; }
    unsigned long getSpeakerNumber() const { return _session_per_speaker.size(); }
    unsigned long getSessionNumber() const { return _n_sessions; }
    unsigned long getSpeakerSessionNumber(unsigned long spk) const { return _session_per_speaker.at(spk); }
    std::vector<double> &getData() { return _data.host(); }   // host view, downloaded on demand
    DVec &deviceData() { return _data; }
    const std::vector<double> &getMean() const { return _mean; }
    const std::vector<double> &getSpeakerMeans() const { return _speaker_means; } // [vectSize x n_speakers]
    void computeAll();                                   // :353-387
    void lengthNorm();                                   // :436-463
    void center(const std::vector<double> &mu);          // :466-474
    void centerPerSpeaker();                             // :488-495 (means are NOT recomputed, like the reference)
    void rotateLeft(const std::vector<double> &M, unsigned long rows); // :498-513, M [rows x vectSize]
    void computeCovMat(std::vector<double> &Sigma, std::vector<double> &W, std::vector<double> &B);   // :527-566
    void computeWccnChol(std::vector<double> &WCCN);     // :1124-1176
    void computeMahalanobis(std::vector<double> &M);     // :1366-1378
    void computeScatterMat(std::vector<double> &SB, std::vector<double> &SW);                          // :1610-1644
    void computeLDA(std::vector<double> &ldaMat, unsigned long ldaRank, bool scatterMatrices = false);  // :1381-1413, [rank x vectSize]
    // :1822-1929: nbIt iterations of {covariances, EFR (Sigma) or sphNorm (W) matrix, center, rotate, lengthNorm};
    // the matrices and means of every iteration are returned instead of being written to files
    void sphericalNuisanceNormalization(unsigned long nbIt, bool sphNorm, std::vector<std::vector<double> > &mats,
                                        std::vector<std::vector<double> > &means);
    // one PldaModel::em_iteration on this set (used by PldaModel; the data is centred by Delta in place)
    void emIteration(unsigned long rankF, unsigned long rankG, std::vector<double> &F, std::vector<double> &G, std::vector<double> &Sigma,
                     std::vector<double> &Delta, const std::vector<int64_t> &sessionsPerSpeaker);
    // :1931-2005: apply stored matrices / means
    void applySphericalNuisanceNormalization(const std::vector<std::vector<double> > &mats, const std::vector<std::vector<double> > &means);

  private:
    std::vector<int64_t> sps64() const;
    GpuServer &_srv;
    unsigned long _vectSize, _n_sessions;
    DVec _data;                                   // [vectSize x n_sessions], device resident
    std::vector<double> _mean, _speaker_means;
    std::vector<unsigned long> _session_per_speaker;
};

// PldaModel in training mode (PldaTools.cpp:2043-2120, 2329-2343, 2790-2815).  The reference draws F and G at
// random when pldaLoadInitMatrices is false (initF / initG, glibc rand()); here they are always given, like its
// pldaLoadInitMatrices branch.
class PldaModel {
  public:
    PldaModel(PldaDev &dev, unsigned long rankF, unsigned long rankG, const std::vector<double> &F, const std::vector<double> &G,
              const std::vector<double> &Sigma);
    void em_iteration();                                        // :2329-2343
    std::vector<double> &getF() { return _F; }                  // [vectSize x rankF]
    std::vector<double> &getG() { return _G; }                  // [vectSize x rankG]
    std::vector<double> &getSigma() { return _Sigma; }          // [vectSize x vectSize]
    std::vector<double> &getDelta() { return _Delta; }          // minimum-divergence mean shift
    const std::vector<double> &getOriginalMean() const { return _originalMean; }
    unsigned long getRankF() const { return _rankF; }
    unsigned long getRankG() const { return _rankG; }

  private:
    PldaDev &_Dev;
    unsigned long _rankF, _rankG, _vectSize;
    std::vector<double> _F, _G, _Sigma, _Delta, _originalMean;
};

// ---- IvTest (LIA_SpkDet/IvTest/src/IvTest.cpp:73-471): normalisation estimated on a development set, applied to the
// enrolment and test vectors, then one of the four scoring rules.  Everything the reference exchanges through matrix
// files between the steps is kept in memory.
struct IvTestCfg {
    bool ivNorm = false;                  // ivNorm
    unsigned long ivNormIterationNb = 1;  // ivNormIterationNb
    bool sphNorm = false;                 // ivNormEfrMode == "sphNorm" (else EFR)
    bool LDA = false;                     // LDA
    unsigned long ldaRank = 0;            // ldaRank
    bool WCCN = false;                    // wccn (cosine scoring only)
    std::string scoring = "cosine";       // cosine | mahalanobis | 2cov | plda
    unsigned long pldaRankF = 0, pldaRankG = 0, pldaNbIt = 0; // pldaEigenVoiceNumber / pldaEigenChannelNumber / pldaNbIt
};
// dev: development set (modified: normalised in place like the reference's PldaDev).
// enrol [dim x nEnrol] one enrolment vector per column, enrolPerModel[m] consecutive columns per model (cosine,
// mahalanobis and 2cov score the MEAN of a model's vectors, plda their sum with the session count, PldaTools.cpp:4206-4221);
// test [dim x nTest].  pldaF / pldaG / pldaSigma: initial matrices of the PLDA EM (ignored for the other rules).
// Returns scores [nModels x nTest].
std::vector<double> ivTest(GpuServer &srv, const IvTestCfg &cfg, PldaDev &dev, std::vector<double> enrol,
                           const std::vector<unsigned long> &enrolPerModel, std::vector<double> test, unsigned long nTest,
                           const std::vector<double> &pldaF, const std::vector<double> &pldaG, const std::vector<double> &pldaSigma);

// TVAcc::computeEigenProblem (AccumulateTVStat.cpp:2997-3102) for the SYMMETRIC matrices it is used on (the weighted
// covariance W): cyclic Jacobi on the host, eigenvalues sorted descending, eigenVect[k*rank + j] = component k of
// the j-th eigenvector (the reference's Eigen / LAPACK solver returns its own column order and sign; any orthonormal
// eigenbasis gives the same approximate i-vectors).  eigenVal [rank] holds the diagonal.
void computeEigenProblem(const std::vector<double> &EP, unsigned long n, std::vector<double> &eigenVect,
                         std::vector<double> &eigenVal, unsigned long rank);

} // namespace liagpu
