// liatools_gpu.cpp -- see liatools_gpu.h.  Host control flow of the LIA_SpkTools hot-path drivers;
// all frame x Gaussian arithmetic happens in libgmmiv (HIP kernels).
#include "liatools_gpu.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>

#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace liagpu {

static void hipcheck(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw Exception(std::string(what) + ": " + hipGetErrorString(e));
}

// LIAGPU_TRACE=1: wall time of the host-side stages of a training iteration on stderr (tools/host_world_time.py reads it)
static bool traceOn()
{
    static const int on = [] { const char *e = getenv("LIAGPU_TRACE"); return (e && *e && *e != '0') ? 1 : 0; }();
    return on != 0;
}
struct StageClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
        return ms;
    }
};

unsigned long totalFrame(const SegCluster &c)
{
    unsigned long n = 0;
    for (const Seg &s : c) n += s.length;
    return n;
}

unsigned long timeToFrameIdx(double time, double frameLength)
{
    // SegTools.cpp:135-142: the quotient is TRUNCATED; only a fractional part above 0.99999 (a time that is a whole
    // number of frames up to rounding, e.g. 0.07 / 0.01 = 6.999999...) moves to the next frame
    const double q = time / frameLength, whole = floor(q);
    return (unsigned long)whole + ((q - whole) > 0.99999 ? 1ul : 0ul);
}
double frameIdxToTime(unsigned long idx, double frameLength)
{
    // SegTools.cpp:143-148: the time in whole milliseconds, truncated
    return (double)(unsigned long)((double)idx * 1000.0 * frameLength) / 1000.0;
}

Seg segFromLabel(double begin_s, double end_s, double frameLength, unsigned long source)
{
    // SegTools.cpp:265-271: both times go through timeToFrameIdx, end frame INCLUSIVE
    Seg s;
    const unsigned long b = timeToFrameIdx(begin_s, frameLength), e = timeToFrameIdx(end_s, frameLength);
    s.begin = b;
    s.length = e - b + 1;
    s.source = source;
    return s;
}

// ---- GpuServer ---------------------------------------------------------------------------------
GpuServer::GpuServer(int device)
{
    if (gmmiv_ctx_create(device, nullptr, &_ctx) != 0) throw Exception(gmmiv_last_error());
}
GpuServer::~GpuServer()
{
    (void)gmmiv_ctx_sync(_ctx);
    for (void *p : _ws) if (p) (void)hipFree(p);
    gmmiv_ctx_destroy(_ctx);
}
void *GpuServer::workspace(int slot, size_t bytes)
{
    if (slot < 0 || slot >= 8) throw Exception("GpuServer::workspace: bad slot");
    if (bytes == 0) bytes = 8;
    if (_wsBytes[slot] < bytes) {
        sync(); // kernels in flight may still use the old block
        if (_ws[slot]) hipcheck(hipFree(_ws[slot]), "GpuServer::workspace: hipFree");
        _ws[slot] = nullptr; _wsBytes[slot] = 0;
        const size_t want = bytes + bytes / 8;
        hipcheck(hipMalloc(&_ws[slot], want), "GpuServer::workspace: hipMalloc");
        _wsBytes[slot] = want;
    }
    return _ws[slot];
}
// DEGENERATE INPUTS (include/gmmiv.h): whether the C ABI may skip its per-call screening pass is decided PER CALL from the
// FeatureBuffer whose frames the call reads -- never as a context-wide setting: a TVAcc / JFAAcc may be handed the frames of a
// buffer that lives on ANOTHER server, and a clean buffer on this server says nothing about those.  A value the
// user set on the context is honoured (assume_finite 1 = "my data are clean") and restored when the call returns or throws.
FiniteScope::FiniteScope(GpuServer &srv, const FeatureBuffer &fs) : _ctx(srv.ctx())
{
    _prev = gmmiv_ctx_set_option(_ctx, "assume_finite", 0);
    if (_prev > 0 || fs.unusableFrames() == 0) (void)gmmiv_ctx_set_option(_ctx, "assume_finite", 1);
}
FiniteScope::~FiniteScope() { (void)gmmiv_ctx_set_option(_ctx, "assume_finite", _prev > 0 ? 1 : 0); }
void GpuServer::check(int rc) const
{
    if (rc != 0) throw Exception(gmmiv_last_error());
}

// ---- DVec --------------------------------------------------------------------------------------
DVec::~DVec()
{
    if (_d) { if (_srv) (void)gmmiv_ctx_sync(_srv->ctx()); (void)hipFree(_d); }
}
void DVec::sync() const
{
    if (_srv) _srv->sync();
    else hipcheck(hipDeviceSynchronize(), "DVec: sync");
}
void *DVec::st() const { return _srv ? _srv->stream() : nullptr; }
void DVec::reserve(size_t n)
{
    if (n <= _cap) return;
    sync(); // kernels in flight may still use the old buffer
    if (_d) hipcheck(hipFree(_d), "DVec: hipFree");
    _d = nullptr; _cap = 0;
    hipcheck(hipMalloc((void **)&_d, (n ? n : 1) * sizeof(double)), "DVec: hipMalloc");
    _cap = n;
}
void DVec::assign(size_t n, double v)
{
    reserve(n);
    _n = n;
    _hostValid = _hostDirty = false;
    _h.clear();
    if (n == 0) return;
    if (v == 0.0) hipcheck(hipMemsetAsync(_d, 0, n * sizeof(double), (hipStream_t)st()), "DVec: hipMemsetAsync");
    else {
        const std::vector<double> tmp(n, v);
        hipcheck(hipMemcpyAsync(_d, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)st()), "DVec: upload");
        sync(); // tmp goes out of scope
    }
}
void DVec::set(const double *h, size_t n)
{
    reserve(n);
    _n = n;
    _hostValid = _hostDirty = false;
    _h.clear();
    if (n == 0) return;
    hipcheck(hipMemcpyAsync(_d, h, n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)st()), "DVec: upload");
    sync(); // the caller may free h on return
}
void DVec::copyFrom(const DVec &o)
{
    const double *src = o.cdev();
    reserve(o._n);
    _n = o._n;
    _hostValid = _hostDirty = false;
    _h.clear();
    if (_n == 0) return;
    hipcheck(hipMemcpyAsync(_d, src, _n * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)st()), "DVec: copy"); // in stream order
}
void DVec::swap(DVec &o)
{
    std::swap(_srv, o._srv); std::swap(_d, o._d); std::swap(_n, o._n); std::swap(_cap, o._cap);
    _h.swap(o._h); std::swap(_hostValid, o._hostValid); std::swap(_hostDirty, o._hostDirty);
}
const double *DVec::cdev() const
{
    if (_hostDirty) {
        if (_h.size() != _n) throw Exception("DVec: the host view was resized");
        if (_n) hipcheck(hipMemcpyAsync(_d, _h.data(), _n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)st()), "DVec: upload");
        sync();
        _hostDirty = false; // _h still mirrors the device copy
    }
    return _d;
}
double *DVec::dev()
{
    (void)cdev();
    _hostValid = false; // a kernel is about to write
    return _d;
}
const std::vector<double> &DVec::chost() const
{
    if (!_hostValid && !_hostDirty) {
        _h.resize(_n);
        if (_n) hipcheck(hipMemcpyAsync(_h.data(), _d, _n * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)st()), "DVec: download");
        sync();
        _hostValid = true;
    }
    return _h;
}
std::vector<double> &DVec::host()
{
    (void)chost();
    _hostDirty = true;
    return _h;
}
void DVec::get(double *h, size_t n, size_t offset) const
{
    if (offset + n > _n) throw Exception("DVec::get: range out of bounds");
    const double *src = cdev();
    if (n) hipcheck(hipMemcpyAsync(h, src + offset, n * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)st()), "DVec: download");
    sync();
}

// ---- FeatureBuffer -----------------------------------------------------------------------------
FeatureBuffer::FeatureBuffer(GpuServer &srv, const float *frames, unsigned long nFrames, unsigned long vectSize,
                             const std::vector<unsigned long> &sourceFirstFrame)
    : _srv(srv), _n(nFrames), _d(vectSize), _first(sourceFirstFrame)
{
    const size_t bytes = (size_t)(nFrames ? nFrames : 1) * vectSize * sizeof(float);
    hipcheck(hipMalloc((void **)&_dev, bytes), "FeatureBuffer: hipMalloc");
    if (nFrames) hipcheck(hipMemcpy(_dev, frames, (size_t)nFrames * vectSize * sizeof(float), hipMemcpyHostToDevice), "FeatureBuffer: upload");
    // screened ONCE here instead of in every call of every iteration
    int64_t bad = 0;
    if (gmmiv_count_unusable_frames(srv.ctx(), _dev, GMMIV_F32, (int64_t)nFrames, (int64_t)vectSize, (int)vectSize, &bad) != 0) {
        (void)hipFree(_dev); // the destructor does not run for a half-built object
        _dev = nullptr;
        throw Exception(gmmiv_last_error());
    }
    _unusable = (unsigned long)bad;
}
FeatureBuffer::~FeatureBuffer()
{
    (void)gmmiv_ctx_sync(_srv.ctx()); // kernels / copies in flight may still use the buffers
    if (_dev) (void)hipFree(_dev);
    if (_sel) (void)hipFree(_sel);
    if (_dRuns) (void)hipFree(_dRuns);
    if (_hRuns) (void)hipHostFree(_hRuns);
    if (_runsCopied) (void)hipEventDestroy((hipEvent_t)_runsCopied);
}

// The cluster as RUNS of adjacent frames (source frame, output row, length) in the pinned table; segments that follow each other
// in the buffer are merged (at baggedFrameProbability 1 the whole cluster becomes ONE run), then cut into pieces of <= 64 frames
// (one wavefront of k_gather_runs moves one piece).  Returns the number of merged runs.
unsigned long FeatureBuffer::buildRuns(const SegCluster &c, unsigned long &nSelected, unsigned long &firstFrame, size_t &nPieces)
{
    const size_t kPiece = 64;
    // worst case: every segment its own run + one extra piece per 64 frames
    size_t need = c.size() + totalFrame(c) / kPiece + 1;
    if (need > _runsCap) {
        if (_runsCopied) hipcheck(hipEventSynchronize((hipEvent_t)_runsCopied), "FeatureBuffer: hipEventSynchronize");
        if (_hRuns) hipcheck(hipHostFree(_hRuns), "FeatureBuffer: hipHostFree");
        _hRuns = nullptr; _runsCap = 0;
        need += need / 4;
        hipcheck(hipHostMalloc((void **)&_hRuns, need * 3 * sizeof(int64_t), hipHostMallocDefault), "FeatureBuffer: hipHostMalloc(runs)");
        _runsCap = need;
    } else if (_runsCopied)
        hipcheck(hipEventSynchronize((hipEvent_t)_runsCopied), "FeatureBuffer: hipEventSynchronize"); // the last upload has read the table
    int64_t *t = _hRuns;
    size_t np = 0;
    unsigned long dst = 0, nMerged = 0, curSrc = 0, curLen = 0;
    firstFrame = 0;
    auto flush = [&]() {
        while (curLen > 0) {
            const unsigned long l = curLen < kPiece ? curLen : kPiece;
            t[3 * np] = (int64_t)curSrc; t[3 * np + 1] = (int64_t)dst; t[3 * np + 2] = (int64_t)l;
            ++np; curSrc += l; dst += l; curLen -= l;
        }
    };
    for (const Seg &s : c) {
        if (s.length == 0) continue;
        const unsigned long b = s.begin + getFirstFeatureIndexOfASource(s.source);
        if (b + s.length > _n) throw Exception("segment ends after the last frame of the feature buffer");
        if (nMerged && b == curSrc + curLen) { curLen += s.length; continue; }
        flush();
        if (!nMerged) firstFrame = b;
        ++nMerged;
        curSrc = b; curLen = s.length;
    }
    flush();
    nSelected = dst;
    nPieces = np;
    return nMerged;
}

const float *FeatureBuffer::select(const SegCluster &c, unsigned long &nSelected)
{
    // the frame list the reference walks with seekFeature / readFeature (AccumulateStat.cpp:121-128), as a device matrix: everything
    // below is ENQUEUED on the context's stream -- the table upload comes from pinned memory -- so the host returns at once and may
    // prepare the next selection while the kernels of this one run
    unsigned long first = 0;
    size_t nPieces = 0;
    const unsigned long nMerged = buildRuns(c, nSelected, first, nPieces);
    if (nSelected == 0) return _dev;
    if (nMerged == 1) return _dev + (size_t)first * _d; // one contiguous run: no copy
    hipStream_t st = (hipStream_t)_srv.stream();
    if (_selCap < nSelected) {
        _srv.sync();
        if (_sel) hipcheck(hipFree(_sel), "FeatureBuffer: hipFree");
        _sel = nullptr; _selCap = 0;
        const unsigned long cap = nSelected + nSelected / 8;
        hipcheck(hipMalloc((void **)&_sel, (size_t)cap * _d * sizeof(float)), "FeatureBuffer: hipMalloc(select)");
        _selCap = cap;
    }
    if (_dRunsCap < nPieces) {
        _srv.sync();
        if (_dRuns) hipcheck(hipFree(_dRuns), "FeatureBuffer: hipFree");
        _dRuns = nullptr; _dRunsCap = 0;
        const size_t cap = nPieces + nPieces / 4;
        hipcheck(hipMalloc((void **)&_dRuns, cap * 3 * sizeof(int64_t)), "FeatureBuffer: hipMalloc(runs)");
        _dRunsCap = cap;
    }
    hipcheck(hipMemcpyAsync(_dRuns, _hRuns, nPieces * 3 * sizeof(int64_t), hipMemcpyHostToDevice, st), "FeatureBuffer: upload(runs)");
    if (!_runsCopied) {
        hipEvent_t e;
        hipcheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "FeatureBuffer: hipEventCreate");
        _runsCopied = e;
    }
    hipcheck(hipEventRecord((hipEvent_t)_runsCopied, st), "FeatureBuffer: hipEventRecord");
    _srv.check(gmmiv_gather_runs(_srv.ctx(), _dev, GMMIV_F32, (int64_t)_d, (int)_d, _dRuns, (int64_t)nPieces, _sel));
    return _sel;
}

// ---- MixtureGD ---------------------------------------------------------------------------------
MixtureGD::MixtureGD(unsigned long distribCount, unsigned long vectSize)
    : _c(distribCount), _d(vectSize), _w(distribCount, 1.0 / distribCount), _mean(distribCount * vectSize, 0.0),
      _cov(distribCount * vectSize, 1.0), _covInv(distribCount * vectSize, 1.0)
{
}
void MixtureGD::computeAll()
{
    for (size_t i = 0; i < _cov.size(); ++i) _covInv[i] = 1.0 / _cov[i];
}

DeviceMixture::DeviceMixture(GpuServer &srv, const MixtureGD &m) : _srv(srv), _c(m.getDistribCount()), _d(m.getVectSize())
{
    MixtureGD &mm = const_cast<MixtureGD &>(m);
    srv.check(gmmiv_gmm_create(srv.ctx(), (int)m.getDistribCount(), (int)m.getVectSize(), mm.weights().data(), mm.means().data(),
                               m.covInvs().data(), &_g));
}
DeviceMixture::~DeviceMixture() { gmmiv_gmm_destroy(_g); }
void DeviceMixture::update(const MixtureGD &m)
{
    MixtureGD &mm = const_cast<MixtureGD &>(m);
    _srv.check(gmmiv_gmm_set(_g, mm.weights().data(), mm.means().data(), m.covInvs().data()));
}

// ---- EMAcc -------------------------------------------------------------------------------------
EMAcc::EMAcc(DeviceMixture &dm, const MixtureGD &model) : _dm(dm), _model(model), _acc(dm.server())
{
    _acc.assign(gmmiv_em_acc_len((int)model.getDistribCount(), (int)model.getVectSize()), 0.0);
}
void EMAcc::resetEM() { _acc.assign(_acc.size(), 0.0); }
double EMAcc::getEMFeatureCount() const
{
    double v = 0.0;
    _acc.get(&v, 1, _acc.size() - 1);
    return v;
}
double EMAcc::getAccumulatedLLK() const
{
    double v = 0.0;
    _acc.get(&v, 1, _acc.size() - 2);
    return v;
}
void EMAcc::addAccEM(const EMAcc &o)
{
    // MixtureStat::addAccEM merges a worker thread's accumulator (AccumulateStat.cpp:286-292): small (2 MB), done on the host view
    const std::vector<double> &b = o._acc.chost();
    std::vector<double> &a = _acc.host();
    for (size_t i = 0; i < a.size(); ++i) a[i] += b[i];
}
MixtureGD EMAcc::getEM() const
{
    MixtureGD out = _model;
    MixtureGD &m = const_cast<MixtureGD &>(_model);
    GpuServer &srv = const_cast<DeviceMixture &>(_dm).server();
    srv.check(gmmiv_em_get(srv.ctx(), (int)m.getDistribCount(), (int)m.getVectSize(), _acc.cdev(), m.means().data(), m.covs().data(),
                           out.weights().data(), out.means().data(), out.covs().data()));
    out.computeAll();
    return out;
}

std::vector<double> FrameAccGD::getMeanVect() const
{
    std::vector<double> m(vectSize);
    for (unsigned long i = 0; i < vectSize; ++i) m[i] = acc[i] / acc[2 * vectSize];
    return m;
}
std::vector<double> FrameAccGD::getCovVect() const
{
    std::vector<double> c(vectSize);
    const double n = acc[2 * vectSize];
    for (unsigned long i = 0; i < vectSize; ++i) {
        const double m = acc[i] / n;
        c[i] = acc[vectSize + i] / n - m * m;
    }
    return c;
}

// ---- AccumulateStat ------------------------------------------------------------------------------
double accumulateStatEM(FeatureBuffer &fs, EMAcc &emAcc, const SegCluster &selectedSegments, double weight)
{
    unsigned long n = 0;
    StageClock clk;
    const float *x = fs.select(selectedSegments, n);
    const double msSel = clk.lap();
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    const double before = emAcc.getAccumulatedLLK();
    const double msBefore = clk.lap();
    srv.check(gmmiv_em_accumulate(srv.ctx(), emAcc.mixture().handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), weight, emAcc.acc().dev()));
    const double msCall = clk.lap();
    const double after = emAcc.getAccumulatedLLK();
    if (traceOn())
        fprintf(stderr, "[liagpu] accumulateStatEM %lu frames: select %.2f ms, llk before %.2f, gmmiv_em_accumulate (enqueue) %.2f, llk after (waits for the kernels) %.2f\n",
                n, msSel, msBefore, msCall, clk.lap());
    return after - before; // weight * sum log lk of this call (AccumulateStat.cpp:143-152)
}
double accumulateStatEM(FeatureBuffer &fs, EMAcc &emAcc, const SegCluster &selectedSegments)
{
    return accumulateStatEM(fs, emAcc, selectedSegments, 1.0);
}

double accumulateStatLLK(FeatureBuffer &fs, DeviceMixture &m, const SegCluster &selectedSegments, double minLLK, double maxLLK)
{
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    double sums[2] = {0.0, 0.0};
    srv.check(gmmiv_llk(srv.ctx(), m.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), minLLK, maxLLK, nullptr, sums));
    return sums[1] > 0 ? sums[0] / sums[1] : 0.0;
}

void accumulateStatLLK(LLKAcc &llkAcc, FeatureBuffer &fs, DeviceMixture &m, const SegCluster &selectedSegments, double weight, double minLLK,
                       double maxLLK)
{
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    double sums[2] = {0.0, 0.0};
    srv.check(gmmiv_llk(srv.ctx(), m.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), minLLK, maxLLK, nullptr, sums));
    llkAcc.sumLLK += weight * sums[0]; // computeAndAccumulateLLK(f, weight): sum of w llk, sum of w
    llkAcc.sumWeight += weight * sums[1];
}
double meanLikelihood(const std::vector<TrainStream> &streams, DeviceMixture &model, double minLLK, double maxLLK)
{
    LLKAcc acc;
    for (const TrainStream &st : streams) accumulateStatLLK(acc, *st.fs, model, *st.segs, 1.0, minLLK, maxLLK);
    return acc.getMeanLLK();
}
double meanLikelihood(const std::vector<TrainStream> &streams, DeviceMixture &model, const std::vector<double> &decision, double minLLK,
                      double maxLLK)
{
    if (decision.size() != streams.size()) throw Exception("meanLikelihood: one decision weight per feature server expected");
    LLKAcc acc;
    for (size_t i = 0; i < streams.size(); ++i) accumulateStatLLK(acc, *streams[i].fs, model, *streams[i].segs, decision[i], minLLK, maxLLK);
    return acc.getMeanLLK();
}

void accumulateStatFrame(FrameAccGD &frameAcc, FeatureBuffer &fs, const SegCluster &selectedSegments)
{
    if (frameAcc.acc.empty()) {
        frameAcc.vectSize = fs.getVectSize();
        frameAcc.acc.assign(2 * frameAcc.vectSize + 1, 0.0);
    }
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    srv.check(gmmiv_frame_moments(srv.ctx(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), (int)fs.getVectSize(), frameAcc.acc.data()));
}

// ---- TrainTools ------------------------------------------------------------------------------------
double setItParameter(double begin, double end, int nbIt, int it)
{
    if (nbIt < 2) return begin;
    const double itVal = (begin - end) / ((double)nbIt - 1);
    return begin - itVal * it;
}

void varianceControl(MixtureGD &model, double flooring, double ceiling, const std::vector<double> &covSignal)
{
    const unsigned long C = model.getDistribCount(), D = model.getVectSize();
    for (unsigned long c = 0; c < C; ++c)
        for (unsigned long v = 0; v < D; ++v) {
            double cov = model.getCov(c, v);
            if (cov <= flooring * covSignal[v]) cov = flooring * covSignal[v];
            if (cov >= ceiling * covSignal[v]) cov = ceiling * covSignal[v];
            model.setCov(c, cov, v);
        }
    model.computeAll();
}

unsigned long computeMeanCov(FeatureBuffer &fs, const SegCluster &seg, std::vector<double> &mean, std::vector<double> &cov)
{
    FrameAccGD acc;
    accumulateStatFrame(acc, fs, seg);
    mean = acc.getMeanVect();
    cov = acc.getCovVect();
    return acc.getCount();
}

unsigned long computeMeanCov(const std::vector<TrainStream> &streams, std::vector<double> &mean, std::vector<double> &cov)
{
    FrameAccGD acc;
    for (const TrainStream &st : streams) accumulateStatFrame(acc, *st.fs, *st.segs);
    mean = acc.getMeanVect();
    cov = acc.getCovVect();
    return acc.getCount();
}

static bool baggedFrame(double p) { return ((double)rand() / (double)RAND_MAX) < p; }

void baggedSegments(const SegCluster &selectedSegments, SegCluster &baggedFrameSegment, double baggedProbability,
                    unsigned long minimumLength, unsigned long maximumLength)
{
    size_t cur = 0;
    bool end = selectedSegments.empty();
    unsigned long beginSeg = 0, lengthSeg = 0;
    if (!end) { beginSeg = selectedSegments[0].begin; lengthSeg = selectedSegments[0].length; }
    while (!end) {
        unsigned long verifyLength = lengthSeg;
        if (verifyLength < minimumLength) verifyLength = minimumLength;
        if (verifyLength > maximumLength) verifyLength = maximumLength;
        bool moveSeg;
        unsigned long length;
        if (lengthSeg <= verifyLength) { moveSeg = true; length = lengthSeg; }
        else { moveSeg = false; length = verifyLength; }
        if (length > 0 && baggedFrame(baggedProbability)) {
            Seg s;
            s.begin = beginSeg; s.length = length; s.source = selectedSegments[cur].source;
            baggedFrameSegment.push_back(s);
        }
        if (moveSeg) {
            ++cur;
            end = cur >= selectedSegments.size();
            if (!end) { beginSeg = selectedSegments[cur].begin; lengthSeg = selectedSegments[cur].length; }
        } else {
            lengthSeg -= length;
            beginSeg += length;
        }
    }
}

void baggedSegments(const SegCluster &selectedSegments, SegCluster &baggedSeg, unsigned long nbBagged, double baggedProbability,
                    unsigned long minimumLength, unsigned long maximumLength)
{
    size_t cur = 0;
    bool end = selectedSegments.empty();
    unsigned long beginSeg = 0, lengthSeg = 0;
    if (!end) { beginSeg = selectedSegments[0].begin; lengthSeg = selectedSegments[0].length; }
    while (!end) {
        unsigned long verifyLength = lengthSeg;
        if (verifyLength < minimumLength) verifyLength = minimumLength;
        if (verifyLength > maximumLength) verifyLength = maximumLength;
        const bool moveSeg = lengthSeg <= verifyLength;
        const unsigned long length = moveSeg ? lengthSeg : verifyLength;
        if (length > 0)
            for (unsigned long idx = 0; idx < nbBagged; ++idx)      // one draw per component for this chunk
                if (baggedFrame(baggedProbability)) {
                    Seg s;
                    s.begin = beginSeg; s.length = length; s.source = selectedSegments[cur].source; s.labelCode = idx;
                    baggedSeg.push_back(s);
                }
        if (moveSeg) {
            ++cur;
            end = cur >= selectedSegments.size();
            if (!end) { beginSeg = selectedSegments[cur].begin; lengthSeg = selectedSegments[cur].length; }
        } else {
            lengthSeg -= length;
            beginSeg += length;
        }
    }
}

// the picked frames of every component (per stream) -> mean; covariance = globalCov; equal weights (TrainTools.cpp:651-658, :745-756)
static void mixtureFromPicks(const std::vector<FeatureBuffer *> &fsTab, const std::vector<std::vector<SegCluster> > &perStreamComponent,
                             MixtureGD &world, const std::vector<double> &globalCov, std::vector<unsigned long> *frameCount)
{
    const unsigned long C = world.getDistribCount(), D = world.getVectSize();
    if (globalCov.size() != D) throw Exception("mixtureInit: globalCov must have vectSize entries");
    if (frameCount) frameCount->assign(C, 0);
    for (unsigned long c = 0; c < C; ++c) {
        FrameAccGD acc; // ONE accumulator per component over all streams (TrainTools.cpp:686-690)
        for (size_t s = 0; s < fsTab.size(); ++s)
            if (!perStreamComponent[s][c].empty()) accumulateStatFrame(acc, *fsTab[s], perStreamComponent[s][c]);
        if (acc.getCount() == 0) {
            char msg[160];
            snprintf(msg, sizeof(msg), "mixtureInit: no frame was picked for component %lu (too few frames for %lu components)", c, C);
            throw Exception(msg);
        }
        if (frameCount) (*frameCount)[c] = acc.getCount();
        const std::vector<double> mean = acc.getMeanVect();
        for (unsigned long i = 0; i < D; ++i) { world.setCov(c, globalCov[i], i); world.setMean(c, mean[i], i); }
        world.weight(c) = 1.0 / (double)C;      // equalizeWeights
    }
    world.computeAll();
}

void mixtureInit(const std::vector<TrainStream> &streams, MixtureGD &world, const std::vector<double> &globalCov, const MixtureInitCfg &cfg,
                 std::vector<unsigned long> *frameCount)
{
    const unsigned long C = world.getDistribCount();
    if (streams.empty()) throw Exception("mixtureInit: no input stream");
    std::vector<FeatureBuffer *> fsTab(streams.size());
    std::vector<std::vector<SegCluster> > picks(streams.size(), std::vector<SegCluster>(C));
    for (size_t stream = 0; stream < streams.size(); ++stream) {
        fsTab[stream] = streams[stream].fs;
        const SegCluster &selectedSegments = *streams[stream].segs;
        const unsigned long total = totalFrame(selectedSegments);
        if (total == 0) throw Exception("mixtureInit: no frame selected");
        // the bagging probability of the stream, folded into several passes when it exceeds 1 (TrainTools.cpp:700-708, as written)
        double proba = (cfg.nbFrameToSelect * streams[stream].weight) / (double)total;
        unsigned long nbIt = 1;
        double tmp = proba;
        while (tmp > 1) {
            ++nbIt;
            tmp /= proba / (double)nbIt;
            // tmp runs 2, 6/p, 24/p^2, ...: for 1 < p < ~4.9 it never returns below 1 and the reference spins forever here
            if (nbIt > 64) throw Exception("mixtureInit: nbFrameToSelect * weight / totalFrame lies in (1, 4.9): the reference's fold of the bagging probability does not terminate for it (TrainTools.cpp:703-706); select fewer frames per component");
        }
        proba = tmp;
        for (unsigned long baggedIt = 0; baggedIt < nbIt; ++baggedIt) {
            SegCluster bagged;
            srand((unsigned)(((stream + 1) * 100) + (baggedIt + 1)));
            baggedSegments(selectedSegments, bagged, C, proba, cfg.baggedMinimalLength, cfg.baggedMaximalLength);
            for (const Seg &s : bagged) picks[stream][s.labelCode].push_back(s); // accumulateStatFrame(*frameAcc[seg->labelCode()], ...)
        }
    }
    mixtureFromPicks(fsTab, picks, world, globalCov, frameCount);
}

void mixtureInit(FeatureBuffer &fs, const SegCluster &selectedSegments, double streamWeight, MixtureGD &world,
                 const std::vector<double> &globalCov, const MixtureInitCfg &cfg, std::vector<unsigned long> *frameCount)
{
    std::vector<TrainStream> one(1);
    one[0].fs = &fs; one[0].segs = &selectedSegments; one[0].weight = streamWeight;
    mixtureInit(one, world, globalCov, cfg, frameCount);
}

void mixtureInitSingleStream(FeatureBuffer &fs, MixtureGD &world, const SegCluster &selectedSegments, const std::vector<double> &globalCov,
                             const MixtureInitCfg &cfg, std::vector<unsigned long> *frameCount)
{
    const unsigned long C = world.getDistribCount();
    double proba = cfg.baggedFrameProbabilityInit / (double)C;
    unsigned long nbIt = 1;
    if (proba > 1) {
        nbIt = (unsigned long)proba + 1;
        proba /= (double)nbIt;
    }
    std::vector<SegCluster> picks(C);
    for (unsigned long baggedIt = 0; baggedIt < nbIt; ++baggedIt)
        for (unsigned long c = 0; c < C; ++c) {
            srand((unsigned)((c + 1) * (baggedIt + 1)));
            baggedSegments(selectedSegments, picks[c], proba, cfg.baggedMinimalLength, cfg.baggedMaximalLength);
        }
    mixtureFromPicks(std::vector<FeatureBuffer *>(1, &fs), std::vector<std::vector<SegCluster> >(1, picks), world, globalCov, frameCount);
}

// ---- component selection / model normalisation (TrainTools.cpp:175-227, :240-315) ---------------------------------
namespace {
struct TabWeightElem { double weight; unsigned long distrib; };
int compF(const void *op1, const void *op2) // GeneralTools.cpp:277-280: never 0 -- the order of equal weights is the C library's
{
    return ((const TabWeightElem *)op1)->weight > ((const TabWeightElem *)op2)->weight ? -1 : 1;
}
}
std::vector<unsigned long> sortByWeight(const MixtureGD &model)
{
    const unsigned long C = model.getDistribCount();
    std::vector<TabWeightElem> tab(C);
    for (unsigned long i = 0; i < C; ++i) { tab[i].weight = model.weight(i); tab[i].distrib = i; }
    qsort(tab.data(), C, sizeof(TabWeightElem), compF); // TabWeight::_sortByWeight (GeneralTools.h:157-164): the same libc call
    std::vector<unsigned long> order(C);
    for (unsigned long i = 0; i < C; ++i) order[i] = tab[i].distrib;
    return order;
}
unsigned long selectComponent(std::vector<bool> &selectCompA, unsigned long nbTop, const MixtureGD &inputM)
{
    const unsigned long C = inputM.getDistribCount();
    if (nbTop > C) throw Exception("selectComponent: nbTop exceeds the number of components");
    selectCompA.assign(C, false);
    const std::vector<unsigned long> order = sortByWeight(inputM);
    for (unsigned long i = 0; i < nbTop; ++i) selectCompA[order[i]] = true;
    return nbTop;
}
unsigned long selectComponent(std::vector<bool> &selectCompA, double wFactor, const MixtureGD &inputM)
{
    const unsigned long C = inputM.getDistribCount();
    selectCompA.assign(C, true);
    unsigned long n = C;
    for (unsigned long i = 0; i < C; ++i)
        if (inputM.weight(i) < wFactor) { selectCompA[i] = false; --n; }
    return n;
}
double reduceModel(const std::vector<bool> &selectCompA, const MixtureGD &inputM, MixtureGD &outputM)
{
    const unsigned long C = inputM.getDistribCount(), D = inputM.getVectSize();
    unsigned long o = 0;
    double totW = 0.0;
    for (unsigned long c = 0; c < C; ++c)
        if (selectCompA[c]) {
            if (o >= outputM.getDistribCount()) throw Exception("reduceModel: the output mixture is too small");
            for (unsigned long i = 0; i < D; ++i) { outputM.setMean(o, inputM.getMean(c, i), i); outputM.setCov(o, inputM.getCov(c, i), i); }
            outputM.weight(o) = inputM.weight(c);
            totW += outputM.weight(o);
            ++o;
        }
    outputM.computeAll();
    return totW;
}
void normalizeWeights(MixtureGD &outputM)
{
    double totW = 0.0;
    for (unsigned long c = 0; c < outputM.getDistribCount(); ++c) totW += outputM.weight(c);
    for (unsigned long c = 0; c < outputM.getDistribCount(); ++c) outputM.weight(c) /= totW;
}
void mixtureFusion(const MixtureGD &mixt, std::vector<double> &mean, std::vector<double> &cov, double &wres)
{
    // the single Gaussian with the mixture's first and second moments, folded in component by component (gaussianFusion)
    const unsigned long C = mixt.getDistribCount(), D = mixt.getVectSize();
    mean.resize(D); cov.resize(D);
    for (unsigned long k = 0; k < D; ++k) { mean[k] = mixt.getMean(0, k); cov[k] = mixt.getCov(0, k); }
    wres = mixt.weight(0);
    double wtmp = wres;
    for (unsigned long i = 1; i < C; ++i) {
        const double w1 = mixt.weight(i), a1 = w1 / (w1 + wtmp), a2 = 1.0 - a1;
        for (unsigned long k = 0; k < D; ++k) {
            const double d = mixt.getMean(i, k) - mean[k];
            cov[k] = a1 * mixt.getCov(i, k) + a2 * cov[k] + a1 * a2 * d * d;
            mean[k] = (a1 * mixt.getMean(i, k)) + (a2 * mean[k]);
        }
        wres = w1 + wtmp;
        wtmp = wres;
    }
}
void normalizeMixture(MixtureGD &mixt, const std::vector<double> &meanSignal, const std::vector<double> &covSignal, bool zeroOne,
                      unsigned long nbIt, bool meanOnly)
{
    const unsigned long C = mixt.getDistribCount(), D = mixt.getVectSize();
    if (!zeroOne && (meanSignal.size() != D || covSignal.size() != D)) throw Exception("normalizeMixture: meanSignal / covSignal must have vectSize entries");
    std::vector<double> tm, tc;
    double wtmp;
    for (unsigned long it = 0; it < nbIt; ++it) {
        mixtureFusion(mixt, tm, tc, wtmp);
        for (unsigned long c = 0; c < C; ++c)
            for (unsigned long i = 0; i < D; ++i) {
                double newMean = mixt.getMean(c, i) - tm[i];
                newMean /= sqrt(tc[i]);
                if (!zeroOne) { newMean *= covSignal[i]; newMean += meanSignal[i]; }
                mixt.setMean(c, newMean, i);
                if (!meanOnly) {
                    double newCov = mixt.getCov(c, i) / tc[i];
                    if (!zeroOne) newCov *= covSignal[i];
                    mixt.setCov(c, newCov, i);
                }
            }
        mixt.computeAll();
    }
}

// ---- trainModelStream ----------------------------------------------------------------------------------------------------
std::vector<double> trainModelStream(const TrainCfg &cfg, FeatureBuffer &fs, const SegCluster &selectedSegments,
                                     const std::vector<double> &globalCov, MixtureGD &world, gmmiv_comm *comm)
{
    return trainModelStream(cfg, fs, selectedSegments, globalCov, world, nullptr, nullptr, comm);
}

std::vector<double> trainModelStream(const TrainCfg &cfg, FeatureBuffer &fs, const SegCluster &selectedSegments,
                                     const std::vector<double> &globalCov, MixtureGD &world, AllReduceFn allReduce, void *user,
                                     gmmiv_comm *comm)
{
    std::vector<TrainStream> one(1);
    one[0].fs = &fs; one[0].segs = &selectedSegments; one[0].weight = 1.0; // reserveMem: 1 / nbStream (TrainWorld.cpp:85)
    return trainModelStream(cfg, one, globalCov, world, allReduce, user, comm);
}

namespace {
// one call of accumulateStatEM in the reference's loop nest: (iteration, stream, bagging pass)
struct EMPass { unsigned long trainIt, stream, baggedIt; double proba; bool lastOfIt; };
}

std::vector<double> trainModelStream(const TrainCfg &cfg, const std::vector<TrainStream> &streams, const std::vector<double> &globalCov,
                                     MixtureGD &world, AllReduceFn allReduce, void *user, gmmiv_comm *comm)
{
    const unsigned long nbStream = streams.size();
    if (nbStream == 0) throw Exception("trainModelStream: no input stream");
    GpuServer &srv = streams[0].fs->server();
    for (const TrainStream &st : streams) {
        if (!st.fs || !st.segs) throw Exception("trainModelStream: NULL stream");
        if (&st.fs->server() != &srv) throw Exception("trainModelStream: every stream must live on the same GpuServer");
        if (st.fs->getVectSize() != world.getVectSize()) throw Exception("trainModelStream: vectSize of a stream differs from the model's");
    }
    const unsigned long initialDistribCount = world.getDistribCount();
    if (cfg.componentReduction && cfg.targetDistribCount > initialDistribCount)
        throw Exception("trainModelStream: targetMixtureDistribCount exceeds the number of components");
    // the reference's loop nest, flattened: the frame selection of a pass depends on seeds only, never on the model, so the bagging
    // of pass k + 1 (rand() over every chunk: ~3 ms per 10^6 frames) runs on the host WHILE the kernels of pass k run on the device
    std::vector<unsigned long> total(nbStream);
    unsigned long nbTotalFrame = 0;
    for (unsigned long s = 0; s < nbStream; ++s) {
        total[s] = totalFrame(*streams[s].segs);
        nbTotalFrame += (unsigned long)((double)total[s] * streams[s].weight);
    }
    const double nbFrameToSelect = cfg.baggedFrameProbability * nbTotalFrame;
    std::vector<EMPass> passes;
    for (unsigned long trainIt = 0; trainIt < cfg.nbTrainIt; ++trainIt) {
        for (unsigned long s = 0; s < nbStream; ++s) {
            unsigned long nbBaggedIt = 1;
            double baggedProba = (nbFrameToSelect * streams[s].weight) / (double)total[s];
            if (baggedProba > 1) {
                nbBaggedIt = (unsigned long)baggedProba + 1;
                baggedProba /= nbBaggedIt;
            }
            for (unsigned long b = 0; b < nbBaggedIt; ++b) passes.push_back(EMPass{trainIt, s, b, baggedProba, false});
        }
        if (!passes.empty() && passes.back().trainIt == trainIt) passes.back().lastOfIt = true;
    }
    struct Prepared { const float *x = nullptr; unsigned long n = 0; double msBag = 0.0, msSel = 0.0; };
    auto prepare = [&](const EMPass &ps) {
        Prepared pr;
        StageClock clk;
        SegCluster baggedFramesCluster;
        baggedFramesCluster.reserve((size_t)((double)total[ps.stream] / (double)std::max<unsigned long>(cfg.baggedMinimalLength, 1) * std::min(ps.proba * 1.05 + 0.01, 1.0)) + 16);
        srand((unsigned)(((ps.trainIt + 1 + cfg.initRand) * 200) + (((ps.stream + 1) * 20) + (ps.baggedIt + 1)))); // TrainTools.cpp:1070
        baggedSegments(*streams[ps.stream].segs, baggedFramesCluster, ps.proba, cfg.baggedMinimalLength, cfg.baggedMaximalLength);
        pr.msBag = clk.lap();
        pr.x = streams[ps.stream].fs->select(baggedFramesCluster, pr.n); // enqueued behind the kernels in flight
        pr.msSel = clk.lap();
        return pr;
    };

    std::vector<double> llkIt;
    // ONE accumulator for all iterations (the reference creates a MixtureStat per iteration; resetEM() is the same state): no device
    // allocation / release (each a device synchronisation) inside the loop -- rebuilt only when componentReduction shrinks the model
    std::unique_ptr<DeviceMixture> dworld(new DeviceMixture(srv, world));
    std::unique_ptr<EMAcc> emAcc(new EMAcc(*dworld, world));
    emAcc->resetEM();
    Prepared cur;
    if (!passes.empty()) cur = prepare(passes[0]);
    double msBag = 0.0, msSel = 0.0, msEnq = 0.0;
    if (cfg.iterationMs) cfg.iterationMs->clear();
    StageClock itClock;
    for (size_t k = 0; k < passes.size(); ++k) {
        const EMPass &ps = passes[k];
        StageClock clk;
        {
            FiniteScope fin(srv, *streams[ps.stream].fs); // screening follows the stream's own buffer
            srv.check(gmmiv_em_accumulate(srv.ctx(), dworld->handle(), cur.x, GMMIV_F32, (int64_t)cur.n, (int64_t)world.getVectSize(), 1.0,
                                          emAcc->acc().dev()));
        }
        msEnq += clk.lap();
        msBag += cur.msBag; msSel += cur.msSel;
        if (k + 1 < passes.size()) cur = prepare(passes[k + 1]);
        if (!ps.lastOfIt) continue;
        const double msPrep = clk.lap();
        const unsigned long trainIt = ps.trainIt;
        const double varianceFlooring = setItParameter(cfg.initVarianceFlooring, cfg.finalVarianceFlooring, (int)cfg.nbTrainIt, (int)trainIt);
        const double varianceCeiling = setItParameter(cfg.initVarianceCeiling, cfg.finalVarianceCeiling, (int)cfg.nbTrainIt, (int)trainIt);
        if (comm) srv.check(gmmiv_allreduce_f64(comm, emAcc->acc().dev(), emAcc->acc().size())); // RCCL on the device accumulator
        else if (allReduce) allReduce(emAcc->flat().data(), emAcc->flat().size(), user);          // sum over ranks == addAccEM over threads
        double tail[2];
        emAcc->acc().get(tail, 2, emAcc->acc().size() - 2); // [sum_t w log lk_t, sum_t w]: waits for the kernels of the iteration
        const double llkPreviousIt = tail[0] / tail[1];
        const double msWait = clk.lap();
        world = emAcc->getEM();
        varianceControl(world, varianceFlooring, varianceCeiling, globalCov);
        bool resized = false;
        if (cfg.componentReduction) { // TrainTools.cpp:1078-1097
            const double diff = (double)(initialDistribCount - cfg.targetDistribCount) / (double)cfg.nbTrainIt;
            unsigned long nbTop = initialDistribCount - (unsigned long)((double)(trainIt + 1) * diff);
            if (trainIt == cfg.nbTrainIt - 1) nbTop = cfg.targetDistribCount;
            if (nbTop < world.getDistribCount()) {
                if (nbTop == 0) throw Exception("trainModelStream: componentReduction leaves no component");
                std::vector<bool> selectCompA;
                const unsigned long nbOutputDistrib = selectComponent(selectCompA, nbTop, world);
                MixtureGD outputM(nbOutputDistrib, world.getVectSize());
                (void)reduceModel(selectCompA, world, outputM);
                normalizeWeights(outputM);
                world = outputM;
                resized = true;
            }
        }
        if (cfg.normalizeModel) // normalizeMixture(*world, trainCfg, config): target N(0, 1)
            normalizeMixture(world, std::vector<double>(), std::vector<double>(), true, cfg.normalizeModelMeanOnly ? cfg.normalizeModelNbIt : 1,
                             cfg.normalizeModelMeanOnly);
        const double msM = clk.lap();
        if (resized) {
            srv.sync();
            emAcc.reset();
            dworld.reset(new DeviceMixture(srv, world));
            emAcc.reset(new EMAcc(*dworld, world));
        } else {
            dworld->update(world);
            emAcc->setModel(world);
        }
        emAcc->resetEM();
        llkIt.push_back(llkPreviousIt);
        if (cfg.iterationMs) { srv.sync(); cfg.iterationMs->push_back(itClock.lap()); }
        if (traceOn())
            fprintf(stderr, "[liagpu] trainModelStream it %lu: host baggedSegments %.2f ms + select %.2f (overlapped with the previous pass), enqueue %.2f, "
                            "next pass prepared in %.2f, wait for the kernels %.2f, getEM + varianceControl %.2f, model update %.2f\n",
                    trainIt, msBag, msSel, msEnq, msPrep, msWait, msM, clk.lap());
        msBag = msSel = msEnq = 0.0;
    }
    return llkIt;
}

// ---- TopGauss ---------------------------------------------------------------------------------------
double TopGauss::compute(DeviceMixture &ubm, FeatureBuffer &fs, const SegCluster &selectedSegments, double topGauss, int topDistribsCount,
                         bool complete, double minLLK, double maxLLK)
{
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    const int cap = (int)std::min<unsigned long>((unsigned long)std::max(topDistribsCount, 1), ubm.getDistribCount());
    std::vector<int32_t> idx((size_t)n * cap), cnt(n);
    std::vector<double> llk(n);
    _nt = n;
    _snsw.assign(n, 0.0); _snsl.assign(n, 0.0);
    int64_t capped = 0;
    srv.check(gmmiv_topgauss_compute(srv.ctx(), ubm.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), cap, topGauss,
                                     complete ? GMMIV_TOP_COMPLETE : GMMIV_TOP_PARTIAL, minLLK, maxLLK, idx.data(), cnt.data(), _snsw.data(),
                                     _snsl.data(), llk.data(), &capped));
    _capped = (unsigned long)capped;
    _nbg.assign(n, 0);
    _nbgcnt = 0;
    for (unsigned long t = 0; t < n; ++t) { _nbg[t] = (unsigned long)cnt[t]; _nbgcnt += _nbg[t]; }
    _idx.clear(); _idx.reserve(_nbgcnt);
    for (unsigned long t = 0; t < n; ++t)
        for (int j = 0; j < cnt[t]; ++j) _idx.push_back((unsigned long)idx[(size_t)t * cap + j]);
    double s = 0.0;
    for (unsigned long t = 0; t < n; ++t) s += llk[t];
    return n ? s / (double)n : 0.0;
}

// ---- EnergyDetector ----------------------------------------------------------------------------------
void energyMixtureInit(MixtureGD &world)
{
    const unsigned long vectSize = world.getVectSize(), distribCount = world.getDistribCount();
    double mean = -2.0;
    const double meanIncrement = distribCount > 1 ? 4.0 / (double)(distribCount - 1) : 1.0;
    for (unsigned long indg = 0; indg < distribCount; ++indg)
        for (unsigned long c = 0; c < vectSize; ++c, mean += meanIncrement) { // the increment sits in the coefficient loop (:175)
            world.setCov(indg, 1.0, c);
            world.setMean(indg, mean, c);
        }
    world.computeAll();
    for (unsigned long indg = 0; indg < distribCount; ++indg) world.weight(indg) = 1.0 / (double)distribCount; // equalizeWeights
}

unsigned long findMaxEnergyDistrib(const MixtureGD &mixt)
{
    unsigned long cmpMax = 0;
    for (unsigned long c = 1; c < mixt.getDistribCount(); ++c)
        if (mixt.getMean(c, 0) > mixt.getMean(cmpMax, 0)) cmpMax = c;
    return cmpMax;
}

unsigned long selectFrames(const std::vector<float> &energy, double threshold, const SegCluster &selectedSeg, SegCluster &outputSeg)
{
    unsigned long countFrames = 0, ind = 0, begin = 0;
    bool in = false;
    for (const Seg &seg : selectedSeg) {
        for (unsigned long idx = seg.begin; idx < seg.begin + seg.length; ++idx) {
            if (idx >= energy.size()) throw Exception("selectFrames: segment beyond the end of the feature file");
            if ((double)energy[idx] > threshold) {
                ++countFrames;
                if (!in) { in = true; begin = ind; }
            } else if (in) {
                in = false;
                Seg s; s.begin = begin; s.length = ind - begin; s.source = seg.source;           // :144 (ind = end + 1)
                outputSeg.push_back(s);
            }
            ++ind;
        }
        if (in) {
            in = false;
            Seg s; s.begin = begin; s.length = ind - begin + 1; s.source = seg.source;           // :151, as written
            outputSeg.push_back(s);
        }
    }
    return countFrames;
}

SegCluster energyDetector(FeatureBuffer &fs, const SegCluster &selectedSegments, const EnergyDetectorCfg &cfg, MixtureGD *energyModelOut,
                          double *thresholdOut)
{
    if (cfg.thresholdMode != "meanStd")
        throw Exception("energyDetector: thresholdMode [" + cfg.thresholdMode + "] needs alize-core's Histo, only meanStd is available");
    GpuServer &srv = fs.server();
    std::vector<double> globalMean, globalCov;
    computeMeanCov(fs, selectedSegments, globalMean, globalCov);                                  // globalMeanCov (:227-231)
    MixtureGD energyModel(cfg.mixtureDistribCount, fs.getVectSize());
    energyMixtureInit(energyModel);
    DeviceMixture dm(srv, energyModel);
    EMAcc emAcc(dm, energyModel);
    for (unsigned long trainIt = 0; trainIt < cfg.nbTrainIt; ++trainIt) {                         // :241-250
        emAcc.resetEM();
        accumulateStatEM(fs, emAcc, selectedSegments);
        energyModel = emAcc.getEM();
        varianceControl(energyModel, cfg.varianceFlooring, cfg.varianceCeiling, globalCov);
        dm.update(energyModel);
        emAcc.setModel(energyModel);
    }
    const unsigned long higher = findMaxEnergyDistrib(energyModel);
    const double threshold = energyModel.getMean(higher, 0) - cfg.alpha * sqrt(energyModel.getCov(higher, 0)); // :274
    // coefficient 0 of every frame, back on the host for the run-length pass
    std::vector<float> energy(fs.getFeatureCount());
    if (!energy.empty()) {
        hipcheck(hipMemcpy2DAsync(energy.data(), sizeof(float), fs.device(), fs.getVectSize() * sizeof(float), sizeof(float), energy.size(),
                                  hipMemcpyDeviceToHost, (hipStream_t)srv.stream()), "energyDetector: download");
        srv.sync();
    }
    SegCluster outputSeg;
    selectFrames(energy, threshold, selectedSegments, outputSeg);
    if (energyModelOut) *energyModelOut = energyModel;
    if (thresholdOut) *thresholdOut = threshold;
    return outputSeg;
}

// ---- GmmTokenizer -------------------------------------------------------------------------------------
// the sorted top list of every selected frame, [n x ctop] (DETERMINE_TOP_DISTRIBS + getTopDistribIndexVector, GmmTokenizer.cpp:71-72, :100-101)
static std::vector<int32_t> topListOfSelection(const SegCluster &selectedSegments, FeatureBuffer &fs, DeviceMixture &world, int ctop,
                                               double minLLK, double maxLLK, unsigned long &n)
{
    if (ctop < 1 || (unsigned long)ctop > world.getDistribCount()) throw Exception("topDistribsCount must lie in 1 .. distribCount");
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs);
    std::vector<int32_t> idx((size_t)n * ctop);
    if (n)
        srv.check(gmmiv_llk_determine_top(srv.ctx(), world.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), ctop, GMMIV_TOP_COMPLETE,
                                          minLLK, maxLLK, idx.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
    return idx;
}

void computeSymbols(const SegCluster &selectedSegments, FeatureBuffer &fs, DeviceMixture &world, std::vector<unsigned long> &stream,
                    int topDistribsCount, double minLLK, double maxLLK)
{
    unsigned long n = 0;
    const std::vector<int32_t> idx = topListOfSelection(selectedSegments, fs, world, topDistribsCount, minLLK, maxLLK, n);
    stream.reserve(stream.size() + n);
    for (unsigned long t = 0; t < n; ++t) stream.push_back((unsigned long)idx[(size_t)t * topDistribsCount]); // v[0].idx (:102-103)
}

void computeConfusionMatrix(const SegCluster &selectedSegments, FeatureBuffer &fs, DeviceMixture &world, unsigned long nBest,
                            std::vector<unsigned long> &mce_matrix, double minLLK, double maxLLK)
{
    const unsigned long C = world.getDistribCount();
    if (mce_matrix.size() != (size_t)C * C) throw Exception("mce_matrix must be distribCount x distribCount");
    unsigned long n = 0;
    const std::vector<int32_t> idx = topListOfSelection(selectedSegments, fs, world, (int)nBest, minLLK, maxLLK, n);
    for (unsigned long t = 0; t < n; ++t) {
        const int32_t *v = &idx[(size_t)t * nBest];
        for (unsigned long i = 0; i < nBest; ++i) mce_matrix[(size_t)v[0] * C + (size_t)v[i]]++; // :73-75
    }
}

double TopGauss::get(DeviceMixture &ubm, FeatureBuffer &fs, const SegCluster &selectedSegments, bool complete, double minLLK, double maxLLK) const
{
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    if (n != _nt) throw Exception("TopGauss::get: the selection holds another number of frames than the stored one");
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    unsigned long cap = 1;
    for (unsigned long t = 0; t < n; ++t) cap = std::max(cap, _nbg[t]);
    std::vector<int32_t> idx((size_t)n * cap, -1); // -1: no entry (never dereferenced by the USE kernel)
    std::vector<double> nllk(n), llk(n);
    unsigned long b = 0;
    for (unsigned long t = 0; t < n; ++t) {
        for (unsigned long j = 0; j < _nbg[t]; ++j) {
            if (_idx[b + j] >= ubm.getDistribCount()) throw Exception("TopGauss::get: stored Gaussian index out of range");
            idx[(size_t)t * cap + j] = (int32_t)_idx[b + j];
        }
        b += _nbg[t];
        nllk[t] = log(_snsl[t]);
    }
    if (b != _nbgcnt) throw Exception("TopGauss::get: idxBegin != _nbgcnt");
    if (n == 0) return 0.0;
    srv.check(gmmiv_llk_use_top(srv.ctx(), ubm.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), (int)cap, idx.data(), nllk.data(),
                                complete ? GMMIV_TOP_COMPLETE : GMMIV_TOP_PARTIAL, minLLK, maxLLK, llk.data()));
    double s = 0.0;
    for (unsigned long t = 0; t < n; ++t) s += llk[t];
    return s / (double)n;
}

unsigned long TopGauss::frameToIdx(unsigned long f) const
{
    unsigned long cnt = 0;
    for (unsigned long t = 0; t < f && t < _nbg.size(); ++t) cnt += _nbg[t];
    return cnt;
}

void TopGauss::write(const std::string &path) const
{
    static_assert(sizeof(unsigned long) == 8, "the reference writes sizeof(unsigned long) bytes per count / index (LP64)");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw Exception("Cannot find nbGaussian file"); // TopGauss.cpp:204
    bool ok = fwrite(&_nt, sizeof(unsigned long), 1, f) == 1 && fwrite(&_nbgcnt, sizeof(unsigned long), 1, f) == 1;
    ok = ok && fwrite(_nbg.data(), sizeof(unsigned long), _nbg.size(), f) == _nbg.size();
    ok = ok && fwrite(_idx.data(), sizeof(unsigned long), _idx.size(), f) == _idx.size();
    ok = ok && fwrite(_snsw.data(), sizeof(double), _snsw.size(), f) == _snsw.size();
    ok = ok && fwrite(_snsl.data(), sizeof(double), _snsl.size(), f) == _snsl.size();
    fclose(f);
    if (!ok) throw Exception("TopGauss::write: short write on " + path);
}

void TopGauss::read(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw Exception("Cannot find nbGaussian file"); // TopGauss.cpp:81
    unsigned long nt = 0, cnt = 0;
    bool ok = fread(&nt, sizeof(unsigned long), 1, f) == 1 && fread(&cnt, sizeof(unsigned long), 1, f) == 1;
    if (ok) {
        // the sizes come from the file: check them against its length before allocating
        long here = ftell(f);
        fseek(f, 0, SEEK_END);
        const long end = ftell(f);
        fseek(f, here, SEEK_SET);
        const unsigned long avail = (unsigned long)(end - here) / 8;
        ok = nt <= avail && cnt <= avail && 3 * nt + cnt == avail;
    }
    if (ok) {
        _nbg.assign(nt, 0); _idx.assign(cnt, 0); _snsw.assign(nt, 0.0); _snsl.assign(nt, 0.0);
        ok = fread(_nbg.data(), sizeof(unsigned long), nt, f) == nt && fread(_idx.data(), sizeof(unsigned long), cnt, f) == cnt &&
             fread(_snsw.data(), sizeof(double), nt, f) == nt && fread(_snsl.data(), sizeof(double), nt, f) == nt;
    }
    fclose(f);
    if (!ok) throw Exception("TopGauss::read: " + path + " is not a complete nbGaussian file");
    _nt = nt; _nbgcnt = cnt; _capped = 0;
    unsigned long s = 0;
    for (unsigned long v : _nbg) s += v;
    if (s != _nbgcnt) throw Exception("TopGauss::read: " + path + ": the per-frame counts do not add up to the stored total");
}

// ---- TrainTarget -------------------------------------------------------------------------------------
void computeMAPOccDep(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg, double frameCount)
{
    const unsigned long C = initModel.getDistribCount(), D = initModel.getVectSize();
    MixtureGD tmp = initModel;
    if (cfg.meanAdapt || cfg.varAdapt)
        for (unsigned long c = 0; c < C; ++c) {
            const double alpha = client.weight(c) * frameCount; // occupation of the component
            if (cfg.meanAdapt) {
                const double a = alpha / (alpha + cfg.meanReg);
                for (unsigned long i = 0; i < D; ++i) tmp.setMean(c, (1 - a) * initModel.getMean(c, i) + a * client.getMean(c, i), i);
            }
            if (cfg.varAdapt) {
                const double a = alpha / (alpha + cfg.varReg);
                for (unsigned long i = 0; i < D; ++i) {
                    const double dm = initModel.getMean(c, i) - client.getMean(c, i);
                    tmp.setCov(c, (1 - a) * initModel.getCov(c, i) + a * client.getCov(c, i) + (1 - a) * a * dm * dm, i);
                }
            }
        }
    if (cfg.weightAdapt) {
        double sum = 0.0;
        for (unsigned long c = 0; c < C; ++c) {
            const double alpha = client.weight(c) * frameCount, a = alpha / (alpha + cfg.weightReg);
            tmp.weight(c) = a * client.weight(c) + (1 - a) * initModel.weight(c);
            sum += tmp.weight(c);
        }
        for (unsigned long c = 0; c < C; ++c) tmp.weight(c) /= sum;
    }
    tmp.computeAll();
    client = tmp;
}

void computeModelBasedMAPOccDep(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg, double frameCount)
{
    computeMAPOccDep(initModel, client, cfg, frameCount); // TrainTools.cpp:491-536 repeats :445-489 statement for statement
}
void computeMAPConst(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg)
{
    const unsigned long C = initModel.getDistribCount(), D = initModel.getVectSize();
    MixtureGD tmp = initModel;
    if (cfg.meanAdapt) {
        const double alpha = cfg.meanAlpha;
        for (unsigned long c = 0; c < C; ++c)
            for (unsigned long i = 0; i < D; ++i) tmp.setMean(c, (alpha * tmp.getMean(c, i)) + ((1 - alpha) * client.getMean(c, i)), i);
    }
    client = tmp; // variances and weights of the init model (the var / weight branches are "TODO" in the reference); no computeAll there either
}
void computeMAPConst2(const MixtureGD &initModel, MixtureGD &client, const MAPCfg &cfg)
{
    const unsigned long C = initModel.getDistribCount(), D = initModel.getVectSize();
    MixtureGD tmp = initModel;
    if (cfg.meanAdapt) {
        const double alpha = cfg.meanAlpha;
        for (unsigned long c = 0; c < C; ++c)
            for (unsigned long i = 0; i < D; ++i) {
                const double res = ((alpha * tmp.weight(c) * tmp.getMean(c, i)) + ((1 - alpha) * client.weight(c) * client.getMean(c, i))) /
                                   (tmp.weight(c) * alpha + client.weight(c) * (1 - alpha));
                tmp.setMean(c, res, i);
            }
    }
    client = tmp;
}
void computeMAP(const MixtureGD &initModel, MixtureGD &client, unsigned long frameCount, const MAPCfg &cfg)
{
    if (cfg.method == "MAPConst") computeMAPConst(initModel, client, cfg);
    else if (cfg.method == "MAPOccDep") computeMAPOccDep(initModel, client, cfg, (double)frameCount);
    else if (cfg.method == "MAPConst2") computeMAPConst2(initModel, client, cfg);
    else if (cfg.method == "MAPModelBased") computeModelBasedMAPOccDep(initModel, client, cfg, (double)frameCount);
    // else: "mapAlgo unknown, No adaptation will be perform" (TrainTools.cpp:555) -- the client keeps its ML estimate
}

void adaptModel(FeatureBuffer &fs, const SegCluster &selectedSegments, const MixtureGD &aprioriModel, MixtureGD &clientMixture,
                const MAPCfg &mapCfg)
{
    DeviceMixture dclient(fs.server(), clientMixture);
    EMAcc emAcc(dclient, clientMixture); // one accumulator for all iterations (see trainModelStream)
    for (unsigned long trainIt = 0; trainIt < mapCfg.nbTrainIt; ++trainIt) {
        SegCluster bagged;
        baggedSegments(selectedSegments, bagged, mapCfg.baggedFrameProbability, 3, 7); // before srand(), as the reference
        emAcc.resetEM();
        srand((unsigned)trainIt);
        accumulateStatEM(fs, emAcc, bagged);
        clientMixture = emAcc.getEM();
        computeMAP(aprioriModel, clientMixture, (unsigned long)emAcc.getEMFeatureCount(), mapCfg);
        if (mapCfg.normalizeModel) // normalizeMixture(clientMixture, mapCfg, config), TrainTools.cpp:898: target N(0, 1)
            normalizeMixture(clientMixture, std::vector<double>(), std::vector<double>(), true,
                             mapCfg.normalizeModelMeanOnly ? mapCfg.normalizeModelNbIt : 1, mapCfg.normalizeModelMeanOnly);
        dclient.update(clientMixture);
        emAcc.setModel(clientMixture);
    }
}

// ---- ComputeTest -------------------------------------------------------------------------------------
std::vector<double> computeTestLLR(FeatureBuffer &fs, const SegCluster &selectedSegments, DeviceMixture &world,
                                   std::vector<DeviceMixture *> &clients, int topDistribsCount, bool complete,
                                   double minLLK, double maxLLK, bool segmentalMode)
{
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    const int mode = complete ? GMMIV_TOP_COMPLETE : GMMIV_TOP_PARTIAL;
    // a model with fewer Gaussians than topDistribsCount selects all of them (the row stride of idx follows the clamped count)
    topDistribsCount = (int)std::min<unsigned long>((unsigned long)std::max(topDistribsCount, 1), world.getDistribCount());
    // the world's top-C' indices and non-top remainder STAY on the device for the client passes (40 bytes per frame that every
    // client would otherwise upload again); only the per-frame log-likelihoods come back for the segment means
    int32_t *idx = (int32_t *)srv.workspace(0, (size_t)n * topDistribsCount * sizeof(int32_t));
    double *nllk = (double *)srv.workspace(1, (size_t)n * sizeof(double));
    // the per-frame log-likelihoods stay on the device as well: row 0 = world, rows 1.. = clients; only the segment means -- the
    // scores ComputeTest prints -- cross PCIe (8 bytes per frame and model did before: 40 MB for 10^6 frames and 4 clients)
    const size_t nRows = 1 + clients.size();
    double *llkw = (double *)srv.workspace(2, (size_t)n * nRows * sizeof(double)), *llkc = llkw + n;
    // world: DETERMINE_TOP_DISTRIBS on every frame (worldDecime = 1)
    srv.check(gmmiv_llk_determine_top(srv.ctx(), world.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), topDistribsCount, mode,
                                      minLLK, maxLLK, idx, nullptr, nullptr, nllk, nullptr, llkw));
    // clients: USE_TOP_DISTRIBS with the world's indices (+ the world's non-top remainder if COMPLETE), all models of the line in ONE call
    // (a launch, a copy back and a synchronisation per client cost more than the kernel on segments of a few thousand frames)
    std::vector<const gmmiv_gmm *> handles(clients.size());
    for (size_t ci = 0; ci < clients.size(); ++ci) handles[ci] = clients[ci]->handle();
    if (!clients.empty())
        srv.check(gmmiv_llk_use_top_multi(srv.ctx(), (int)handles.size(), handles.data(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(),
                                          topDistribsCount, idx, nllk, mode, minLLK, maxLLK, llkc));
    const size_t nseg = segmentalMode ? selectedSegments.size() : 1;
    std::vector<int64_t> segBegin(nseg + 1, 0);
    if (segmentalMode)
        for (size_t s = 0; s < nseg; ++s) segBegin[s + 1] = segBegin[s] + (int64_t)selectedSegments[s].length;
    else
        segBegin[1] = (int64_t)n;
    std::vector<double> means(nRows * nseg, 0.0); // [row][segment]
    srv.check(gmmiv_segment_means(srv.ctx(), llkw, (int64_t)n, (int)nRows, segBegin.data(), (int64_t)nseg, means.data()));
    std::vector<double> out(nseg * clients.size(), 0.0);
    for (size_t ci = 0; ci < clients.size(); ++ci)
        for (size_t s = 0; s < nseg; ++s) out[s * clients.size() + ci] = means[(1 + ci) * nseg + s] - means[s];
    return out;
}

WindowLLR::WindowLLR(unsigned long size, unsigned long dec, unsigned long nClient)
    : _size(size), _dec(dec), _nClient(nClient), _bIdx(0), _count(0), _idx(size, 0), _acc(nClient, 0.0), _llr(size * nClient, 0.0) {}
void WindowLLR::dec(unsigned long idxFrame)
{
    if (_count < _size) { // window is not full
        _count++;
        _idx[(_bIdx + _count - 1) % _size] = idxFrame;
    } else {              // full: shift by _dec frames
        for (unsigned long w = 0; w < _dec; ++w) {
            for (unsigned long c = 0; c < _nClient; ++c) _acc[c] -= _llr[_bIdx * _nClient + c];
            _bIdx = (_bIdx + 1) % _size;
        }
        _count -= (_dec - 1);
        _idx[(_bIdx + _count - 1) % _size] = idxFrame;
    }
}
void WindowLLR::accLLR(unsigned long clientIdx, double llr)
{
    _llr[((_bIdx + _count - 1) % _size) * _nClient + clientIdx] = llr;
    _acc[clientIdx] += llr;
}

std::vector<double> computeTestLLR(FeatureBuffer &fs, const SegCluster &selectedSegments, DeviceMixture &world,
                                   std::vector<DeviceMixture *> &clients, int topDistribsCount, bool complete, double minLLK,
                                   double maxLLK, bool segmentalMode, unsigned long worldDecime, unsigned long windowSize,
                                   unsigned long windowDec, std::vector<WindowOut> *windows)
{
    if (worldDecime == 0) throw Exception("computeTestLLR: worldDecime must be >= 1");
    unsigned long n = 0;
    const float *x = fs.select(selectedSegments, n);
    GpuServer &srv = fs.server();
    FiniteScope fin(srv, fs); // screening of THIS call follows THIS buffer
    const int mode = complete ? GMMIV_TOP_COMPLETE : GMMIV_TOP_PARTIAL;
    const int ctop = (int)std::min<unsigned long>((unsigned long)std::max(topDistribsCount, 1), world.getDistribCount());
    std::vector<int32_t> idx((size_t)n * ctop);
    std::vector<double> nllk(n), llkw(n), llkc(n);
    // DETERMINE_TOP_DISTRIBS on every frame; frames that are not a multiple of worldDecime inside their segment then
    // take the top set (and the non-top remainder) of the last frame that is, and the world is re-scored on it
    srv.check(gmmiv_llk_determine_top(srv.ctx(), world.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), ctop, mode, minLLK, maxLLK,
                                      idx.data(), nullptr, nullptr, nllk.data(), nullptr, llkw.data()));
    std::vector<char> determined(n, 1);
    std::vector<unsigned long> frameIdx(n); // absolute frame index (window bookkeeping)
    {
        size_t t = 0;
        for (const Seg &sg : selectedSegments) {
            const unsigned long first = sg.begin + fs.getFirstFeatureIndexOfASource(sg.source);
            size_t last = t;
            for (unsigned long f = 0; f < sg.length; ++f, ++t) {
                frameIdx[t] = first + f;
                if (f % worldDecime == 0) last = t;
                else {
                    determined[t] = 0;
                    memcpy(&idx[t * ctop], &idx[last * ctop], ctop * sizeof(int32_t));
                    nllk[t] = nllk[last];
                }
            }
        }
    }
    if (worldDecime > 1) {
        std::vector<double> w2(n);
        srv.check(gmmiv_llk_use_top(srv.ctx(), world.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), ctop, idx.data(), nllk.data(),
                                    mode, minLLK, maxLLK, w2.data()));
        for (size_t t = 0; t < n; ++t) if (!determined[t]) llkw[t] = w2[t];
    }
    const size_t nseg = segmentalMode ? selectedSegments.size() : 1, nc = clients.size();
    std::vector<double> out(nseg * nc, 0.0), llkcAll(windowSize ? n * nc : 0);
    auto meanOver = [&](const std::vector<double> &v, size_t b, size_t e) {
        double s = 0.0;
        for (size_t i = b; i < e; ++i) s += v[i];
        return e > b ? s / (double)(e - b) : 0.0;
    };
    for (size_t ci = 0; ci < nc; ++ci) {
        srv.check(gmmiv_llk_use_top(srv.ctx(), clients[ci]->handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), ctop, idx.data(),
                                    nllk.data(), mode, minLLK, maxLLK, llkc.data()));
        size_t off = 0;
        for (size_t s = 0; s < nseg; ++s) {
            const size_t len = segmentalMode ? selectedSegments[s].length : n;
            out[s * nc + ci] = meanOver(llkc, off, off + len) - meanOver(llkw, off, off + len);
            off += len;
        }
        if (windowSize) for (size_t t = 0; t < n; ++t) llkcAll[t * nc + ci] = llkc[t];
    }
    if (windowSize && windows) {
        // `double llkw = 0` is re-initialised for every frame in the reference loop (:160-161), so a frame whose
        // world score was not recomputed contributes llkc - 0 to the window
        windows->clear();
        WindowLLR win(windowSize, windowDec ? windowDec : windowSize, nc);
        for (size_t t = 0; t < n; ++t) {
            win.dec(frameIdx[t]);
            for (size_t ci = 0; ci < nc; ++ci) win.accLLR(ci, llkcAll[t * nc + ci] - (determined[t] ? llkw[t] : 0.0));
            if (win.isEnd()) {
                WindowOut o;
                o.idxBegin = win.getIdxBegin(); o.idxEnd = win.getIdxEnd();
                for (size_t ci = 0; ci < nc; ++ci) o.llr.push_back(win.getLLR(ci));
                windows->push_back(o);
            }
        }
    }
    return out;
}

// ---- TVAcc -----------------------------------------------------------------------------------------
TVAcc::TVAcc(GpuServer &srv, const MixtureGD &ubm, unsigned long rankT, unsigned long nSpeakers)
    : _srv(srv), _ubm(ubm), _dubm(srv, ubm), _rankT(rankT), _n_speakers(nSpeakers), _n_distrib(ubm.getDistribCount()),
      _vectSize(ubm.getVectSize()), _svSize(ubm.getDistribCount() * ubm.getVectSize())
{
    for (DVec *v : {&_ubm_means, &_ubm_invvar, &_statN, &_statF, &_cN, &_cF, &_T, &_W, &_TETt, &_A, &_Cmx, &_R, &_r, &_meanW, &_aMine, &_tAll, &_tMine}) v->bind(srv);
    _n_sessions_global = _n_speakers;
    _ubm_means.set(_ubm.means());       // supervector of means / inverse variances (AccumulateTVStat.cpp:154-162)
    _ubm_invvar.set(_ubm.covInvs());
    _statN.assign(_n_speakers * _n_distrib, 0.0);
    _statF.assign(_n_speakers * _svSize, 0.0);
    _T.assign(_rankT * _svSize, 0.0);
    _W.assign(_n_speakers * _rankT, 0.0);
    _TETt.assign(_n_distrib * gmmiv_tv_packed_len((int)_rankT), 0.0);
    resetTmpAcc();
}

void TVAcc::resetTmpAcc()
{
    // with an overlap communicator A lives in the reduce-scatter's send buffer: equal blocks of Gaussians, the last one zero padded
    const size_t world = _ovComm ? (size_t)gmmiv_comm_world(_ovComm) : 1;
    const size_t Cpad = (_n_distrib + world - 1) / world * world;
    _A.assign(Cpad * gmmiv_tv_packed_len((int)_rankT), 0.0);
    _Cmx.assign(_rankT * _svSize, 0.0);
    _R.assign(_rankT * _rankT, 0.0);
    _r.assign(_rankT, 0.0);
    _meanW.assign(_rankT, 0.0);
}

void TVAcc::loadT(const std::vector<double> &T)
{
    if (T.size() != _rankT * _svSize) throw Exception("loadT: T must be rankT x (distribCount * vectSize)");
    _T.set(T);
}
void TVAcc::setStats(const std::vector<double> &N, const std::vector<double> &F)
{
    if (N.size() != _n_speakers * _n_distrib || F.size() != _n_speakers * _svSize) throw Exception("TVAcc::setStats: dimension mismatch");
    setStats(N.data(), F.data());
}
void TVAcc::setStats(const double *N, const double *F)
{
    _statN.set(N, _n_speakers * _n_distrib);
    _statF.set(F, _n_speakers * _svSize);
}
void TVAcc::storeStats() { _cN.copyFrom(_statN); _cF.copyFrom(_statF); }
void TVAcc::restoreStats()
{
    if (_cN.size() != _statN.size() || _cF.size() != _statF.size()) throw Exception("TVAcc::restoreStats: nothing stored");
    _statN.copyFrom(_cN); _statF.copyFrom(_cF);
}
// restoreStats() + substractM() of an iteration (TotalVariability.cpp:123-124) in one pass over F: F = stored F - N m
void TVAcc::restoreStatsAndSubstractM()
{
    if (_cN.size() != _statN.size() || _cF.size() != _statF.size()) throw Exception("TVAcc::restoreStatsAndSubstractM: nothing stored");
    _statN.copyFrom(_cN);
    _srv.check(gmmiv_tv_subtract_m_to(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _statN.cdev(), _cF.cdev(), _statF.dev(),
                                      _ubm_means.cdev()));
}

void TVAcc::setOverlap(gmmiv_comm *comm)
{
    finishT();
    _ovComm = (comm && gmmiv_comm_world(comm) > 1) ? comm : nullptr;
    _aBegun = false;
}
void TVAcc::hookAReady(void *self)
{
    TVAcc *t = (TVAcc *)self;
    const size_t P = gmmiv_tv_packed_len((int)t->_rankT), world = (size_t)gmmiv_comm_world(t->_ovComm), Cb = (t->_n_distrib + world - 1) / world;
    // an error here cannot leave through the C frames of libgmmiv: it is kept in gmmiv_last_error() and the begin is retried
    // in the serial position by updateTestimate
    t->_aBegun = gmmiv_reduce_scatter_f64_begin(t->_ovComm, t->_A.cdev(), t->_aMine.dev(), Cb * P) == 0;
}
void TVAcc::gatherIntoT()
{
    const int world = gmmiv_comm_world(_ovComm);
    const size_t C = _n_distrib, D = _vectSize, R = _rankT, Cb = (C + world - 1) / world;
    hipStream_t st = (hipStream_t)_srv.stream();
    _srv.check(gmmiv_comm_join(_ovComm));
    for (int g = 0; g < world; ++g) {
        const size_t g0 = g * Cb, gb = g0 >= C ? 0 : std::min(Cb, C - g0);
        if (gb) hipcheck(hipMemcpy2DAsync(_T.dev() + g0 * D, C * D * 8, _tAll.cdev() + (size_t)g * R * Cb * D, Cb * D * 8, gb * D * 8, R, hipMemcpyDeviceToDevice, st),
                         "updateTestimate: gather T");
    }
    _tPending = false;
}
void TVAcc::hookMdFactored(void *self)
{
    TVAcc *t = (TVAcc *)self;
    try { if (t->_tPending) t->gatherIntoT(); } catch (...) { /* reported by finishT() right after the call */ }
}
void TVAcc::finishT()
{
    if (_tPending) gatherIntoT();
}

// updateTestimate on utterance-sharded statistics: rank g owns the Gaussians [g Cb, (g + 1) Cb), Cb = ceil(C / world).
void TVAcc::updateTestimate(gmmiv_comm *comm, unsigned long nSessionsAllRanks)
{
    _n_sessions_global = nSessionsAllRanks;
    const int world = comm ? gmmiv_comm_world(comm) : 1, rank = comm ? gmmiv_comm_rank(comm) : 0;
    if (world <= 1) { updateTestimate(); return; }
    const bool ov = _ovComm != nullptr && _ovComm == comm; // overlapped order: every collective of the step on the side stream
    const size_t P = gmmiv_tv_packed_len((int)_rankT), C = _n_distrib, D = _vectSize, R = _rankT;
    const size_t Cb = (C + world - 1) / world, Cpad = Cb * world;
    // the small sums in one all-reduce: [R x R | R | R]
    DVec small(_srv);
    small.assign(R * R + 2 * R, 0.0);
    hipStream_t st = (hipStream_t)_srv.stream();
    hipcheck(hipMemcpyAsync(small.dev(), _R.cdev(), R * R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: pack");
    hipcheck(hipMemcpyAsync(small.dev() + R * R, _r.cdev(), R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: pack");
    hipcheck(hipMemcpyAsync(small.dev() + R * R + R, _meanW.cdev(), R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: pack");
    DVec aSend(_srv), cSend(_srv), aMineLocal(_srv), cMine(_srv), tMineLocal(_srv), tAllLocal(_srv);
    DVec &aMine = ov ? _aMine : aMineLocal, &tAll = ov ? _tAll : tAllLocal, &tMine = ov ? _tMine : tMineLocal;
    if (ov) {
        if (!_aBegun) { // the hook did not run (host accumulators) or failed: begin A now, in its serial position
            aMine.assign(Cb * P, 0.0);
            _srv.check(gmmiv_reduce_scatter_f64_begin(comm, _A.cdev(), aMine.dev(), Cb * P));
        }
        _aBegun = false;
        _srv.check(gmmiv_allreduce_f64_begin(comm, small.dev(), small.size()));
    } else _srv.check(gmmiv_allreduce_f64(comm, small.dev(), small.size()));
    // A: row blocks are contiguous (padded with empty Gaussians when C does not divide); Cmx: [world][R][Cb D]
    const double *aSrc = _A.cdev();
    if (Cpad != C && !ov) {
        aSend.assign(Cpad * P, 0.0);
        hipcheck(hipMemcpyAsync(aSend.dev(), _A.cdev(), C * P * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: pad A");
        aSrc = aSend.cdev();
    }
    cSend.assign((size_t)world * R * Cb * D, 0.0);
    for (int g = 0; g < world; ++g) {
        const size_t c0 = g * Cb, cb = c0 >= C ? 0 : std::min(Cb, C - c0);
        if (cb) hipcheck(hipMemcpy2DAsync(cSend.dev() + (size_t)g * R * Cb * D, Cb * D * 8, _Cmx.cdev() + c0 * D, C * D * 8, cb * D * 8, R, hipMemcpyDeviceToDevice, st),
                         "updateTestimate: block Cmx");
    }
    cMine.assign(R * Cb * D, 0.0);
    if (ov) {
        _srv.check(gmmiv_reduce_scatter_f64_begin(comm, cSend.cdev(), cMine.dev(), R * Cb * D));
        _srv.check(gmmiv_comm_join(comm));
    } else {
        aMine.assign(Cb * P, 0.0);
        _srv.check(gmmiv_reduce_scatter_f64(comm, aSrc, aMine.dev(), Cb * P));
        _srv.check(gmmiv_reduce_scatter_f64(comm, cSend.cdev(), cMine.dev(), R * Cb * D));
    }
    hipcheck(hipMemcpyAsync(_R.dev(), small.cdev(), R * R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: unpack");
    hipcheck(hipMemcpyAsync(_r.dev(), small.cdev() + R * R, R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: unpack");
    hipcheck(hipMemcpyAsync(_meanW.dev(), small.cdev() + R * R + R, R * 8, hipMemcpyDeviceToDevice, st), "updateTestimate: unpack");
    // the rank's own Gaussians
    const size_t c0 = (size_t)rank * Cb, cb = c0 >= C ? 0 : std::min(Cb, C - c0);
    tMine.assign(R * Cb * D, 0.0);
    if (cb == Cb) _srv.check(gmmiv_tv_update_t(_srv.ctx(), (int)cb, (int)D, (int)R, aMine.cdev(), cMine.cdev(), tMine.dev()));
    else if (cb) { // a cut block: repack its Cmx columns to the width the ABI expects, solve, spread back
        DVec cc(_srv), tt(_srv);
        cc.assign(R * cb * D, 0.0); tt.assign(R * cb * D, 0.0);
            hipcheck(hipMemcpy2DAsync(cc.dev(), cb * D * 8, cMine.cdev(), Cb * D * 8, cb * D * 8, R, hipMemcpyDeviceToDevice, st), "updateTestimate: cut block");
        _srv.check(gmmiv_tv_update_t(_srv.ctx(), (int)cb, (int)D, (int)R, aMine.cdev(), cc.cdev(), tt.dev()));
            hipcheck(hipMemcpy2DAsync(tMine.dev(), Cb * D * 8, tt.cdev(), cb * D * 8, cb * D * 8, R, hipMemcpyDeviceToDevice, st), "updateTestimate: cut block");
    }
    tAll.assign((size_t)world * R * Cb * D, 0.0);
    if (ov) { // T stays in flight: minDivergence() joins it once R is factored (finishT() for a caller that skips it)
        _srv.check(gmmiv_allgather_f64_begin(comm, tMine.cdev(), tAll.dev(), R * Cb * D)); // tMine = _tMine: it outlives this call
        _tPending = true;
        return;
    }
    _srv.check(gmmiv_allgather_f64(comm, tMine.cdev(), tAll.dev(), R * Cb * D));
    for (int g = 0; g < world; ++g) {
        const size_t g0 = g * Cb, gb = g0 >= C ? 0 : std::min(Cb, C - g0);
        if (gb) hipcheck(hipMemcpy2DAsync(_T.dev() + g0 * D, C * D * 8, tAll.cdev() + (size_t)g * R * Cb * D, Cb * D * 8, gb * D * 8, R, hipMemcpyDeviceToDevice, st),
                         "updateTestimate: gather T");
    }
}

void TVAcc::computeAndAccumulateTVStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerLine)
{
    if (segsPerLine.size() != _n_speakers) throw Exception("computeAndAccumulateTVStat: one SegCluster per ndx line expected");
    // all selected frames, line after line, then one batched call (rows are overwritten)
    SegCluster all;
    std::vector<int64_t> uttBegin(_n_speakers + 1, 0);
    for (unsigned long u = 0; u < _n_speakers; ++u) {
        for (const Seg &s : segsPerLine[u]) all.push_back(s);
        uttBegin[u + 1] = uttBegin[u] + (int64_t)totalFrame(segsPerLine[u]);
    }
    unsigned long n = 0;
    const float *x = fs.select(all, n);
    if (&fs.server() != &_srv) fs.server().sync(); // the selection is enqueued on the feature buffer's stream
    FiniteScope fin(_srv, fs); // the frames may come from another server's buffer: the decision follows the buffer, not this context
    _srv.check(gmmiv_tv_stats(_srv.ctx(), _dubm.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), uttBegin.data(),
                              (int64_t)_n_speakers, _statN.dev(), _statF.dev()));
}

void TVAcc::computeAndAccumulateTVStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerFile,
                                       const std::vector<std::vector<unsigned long> > &filesOfLine)
{
    if (filesOfLine.size() != _n_speakers) throw Exception("computeAndAccumulateTVStat: one file list per ndx line expected");
    SegCluster all;
    std::vector<int64_t> fileBegin(segsPerFile.size() + 1, 0), lineOff(_n_speakers + 1, 0), lineFiles;
    for (size_t f = 0; f < segsPerFile.size(); ++f) {
        for (const Seg &s : segsPerFile[f]) all.push_back(s);
        fileBegin[f + 1] = fileBegin[f] + (int64_t)totalFrame(segsPerFile[f]);
    }
    for (unsigned long l = 0; l < _n_speakers; ++l) {
        for (unsigned long f : filesOfLine[l]) {
            if (f >= segsPerFile.size()) throw Exception("computeAndAccumulateTVStat: file index out of range");
            lineFiles.push_back((int64_t)f);
        }
        lineOff[l + 1] = (int64_t)lineFiles.size();
    }
    if (lineFiles.empty()) lineFiles.push_back(0);
    unsigned long n = 0;
    const float *x = fs.select(all, n);
    if (&fs.server() != &_srv) fs.server().sync(); // the selection is enqueued on the feature buffer's stream
    FiniteScope fin(_srv, fs); // the frames may come from another server's buffer: the decision follows the buffer, not this context
    _srv.check(gmmiv_tv_stats_lines(_srv.ctx(), _dubm.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), fileBegin.data(),
                                    (int64_t)segsPerFile.size(), (int64_t)_n_speakers, lineOff.data(), lineFiles.data(), _statN.dev(), _statF.dev()));
}

// ScoreWarp.cpp:68-81: the generator keeps the previous uniform draw as the phase of the next sample
static double g_bm_x1 = 0.0, g_bm_x2 = 0.0;
static void boxMullerGeneratorInit() { g_bm_x1 = (rand() / (float)RAND_MAX); }
static double boxMullerGenerator(double mean, double cov)
{
    g_bm_x2 = g_bm_x1;
    g_bm_x1 = (rand() / (float)RAND_MAX);
    const double y = sqrt(-2.0 * log(g_bm_x1)) * cos(3.14159265358979323846 * 2 * g_bm_x2);
    return y * cov + mean;
}
void TVAcc::initT(const std::string &randomInitLaw)
{
    if (randomInitLaw != "normal") throw Exception("Selected random initialization law does not exist"); // AccumulateTVStat.cpp:750
    const std::vector<double> &iv = _ubm_invvar.chost();
    double norm = 0.0;
    for (unsigned long k = 0; k < _svSize; ++k) norm += iv[k];
    std::vector<double> T(_rankT * _svSize);
    boxMullerGeneratorInit();
    for (size_t e = 0; e < T.size(); ++e) { // row-major (i, j) order like the reference's double loop
        double val = boxMullerGenerator(0.0, 1.0);
        while (std::isnan(val) || std::isinf(val)) val = boxMullerGenerator(0.0, 1.0);
        T[e] = val * norm * 0.001;
    }
    _T.set(T);
}

void TVAcc::substractM()
{
    _srv.check(gmmiv_tv_subtract_m(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _statN.cdev(), _statF.dev(), _ubm_means.cdev()));
}
void TVAcc::estimateTETt()
{
    _srv.check(gmmiv_tv_tett(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, _T.cdev(), _ubm_invvar.cdev(), _TETt.dev()));
}
void TVAcc::estimateW()
{
    _srv.check(gmmiv_tv_estimate_w(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankT, _statN.cdev(), _statF.cdev(),
                                   _T.cdev(), _ubm_invvar.cdev(), _TETt.cdev(), _W.dev()));
}
void TVAcc::estimateAandC()
{
    finishT();
    resetTmpAcc(); // the reference zeroes A, C, R, r, meanW at entry (:1712-1722)
    if (_ovComm) { // the reduce-scatter of A starts from inside the call, as soon as A is complete (under the Cmx GEMM)
        const size_t world = (size_t)gmmiv_comm_world(_ovComm), Cb = (_n_distrib + world - 1) / world;
        _aMine.assign(Cb * gmmiv_tv_packed_len((int)_rankT), 0.0);
        _aBegun = false;
        (void)_A.dev(); (void)_aMine.dev(); // both resident before the hook hands them to the collective
        gmmiv_ctx_set_hook(_srv.ctx(), "tv_a_ready", &TVAcc::hookAReady, this);
    }
    const int rc = gmmiv_tv_estimate_a_and_c(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankT, _statN.cdev(),
                                             _statF.cdev(), _T.cdev(), _ubm_invvar.cdev(), _TETt.cdev(), _W.dev(), _A.dev(), _Cmx.dev(),
                                             _R.dev(), _r.dev(), _meanW.dev());
    if (_ovComm) gmmiv_ctx_set_hook(_srv.ctx(), "tv_a_ready", nullptr, nullptr);
    _srv.check(rc);
    // _meanW stays the SUM of the i-vectors on the device (the all-reduce payload of the sharded form); minDivergence divides (:1791-1794)
}
void TVAcc::updateTestimate()
{
    _srv.check(gmmiv_tv_update_t(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, _A.cdev(), _Cmx.cdev(), _T.dev()));
}
void TVAcc::minDivergence()
{
    // _n_sessions == number of statistics rows in TotalVariability (one session per line)
    std::vector<double> mw = _meanW.chost();                      // meanW /= n (:1791-1794), R floats
    for (double &v : mw) v /= (double)_n_sessions_global;
    if (_tPending) gmmiv_ctx_set_hook(_srv.ctx(), "md_factored", &TVAcc::hookMdFactored, this); // T is still arriving: joined once R is factored
    double *Td = _T.dev();
    const int rc = gmmiv_tv_min_divergence(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, (double)_n_sessions_global, _R.dev(), _r.dev(),
                                           mw.data(), _ubm_means.dev(), Td);
    gmmiv_ctx_set_hook(_srv.ctx(), "md_factored", nullptr, nullptr);
    _srv.check(rc);
    finishT(); // no-op unless the hook could not run
}

void TVAcc::orthonormalizeT()
{
    _srv.check(gmmiv_tv_orthonormalize_t(_srv.ctx(), (int)_rankT, (int64_t)_svSize, _T.dev()));
}

void TVAcc::normStatistics()
{
    _srv.check(gmmiv_tv_norm_statistics(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _statN.cdev(), _statF.dev(),
                                        _ubm_means.cdev(), _ubm_invvar.cdev()));
}
void TVAcc::substractMplusTW()
{
    _srv.check(gmmiv_tv_subtract_m_plus_tw(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankT, _statN.cdev(),
                                           _statF.dev(), _ubm_means.cdev(), _T.cdev(), _W.cdev()));
}
void TVAcc::getMplusTW(std::vector<double> &Sp, const std::vector<unsigned long> &rows)
{
    // Sp[k] = m + T^T w_rows[k] for all k in one device pass: gmmiv_jfa_subtract computes F -= N (m + W T); with N = -1 and
    // F = 0 the result IS the supervector (the same entry point serves JFAAcc::getMplusVY...)
    const size_t n = rows.size();
    Sp.assign(n * _svSize, 0.0);
    if (!n) return;
    std::vector<double> Wsel(n * _rankT);
    const std::vector<double> &W = _W.chost();
    for (size_t k = 0; k < n; ++k) {
        if (rows[k] >= _n_speakers) throw Exception("getMplusTW: row out of range");
        memcpy(&Wsel[k * _rankT], &W[rows[k] * _rankT], _rankT * sizeof(double));
    }
    const std::vector<double> minus1(n * _n_distrib, -1.0);
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)n, (int)_n_distrib, (int)_vectSize, minus1.data(), Sp.data(), nullptr, (int64_t)n,
                                  _ubm_means.cdev(), (int)_rankT, _T.cdev(), Wsel.data(), nullptr, nullptr));
}
void TVAcc::getSpeakerModel(MixtureGD &mixture, unsigned long spk)
{
    if (mixture.getDistribCount() != _n_distrib || mixture.getVectSize() != _vectSize) throw Exception("getSpeakerModel: model shape differs from the UBM's");
    std::vector<double> Sp;
    getMplusTW(Sp, std::vector<unsigned long>(1, spk));
    for (unsigned long c = 0; c < _n_distrib; ++c)              // svToModel: the means, nothing else
        for (unsigned long i = 0; i < _vectSize; ++i) mixture.setMean(c, Sp[c * _vectSize + i], i);
}
double TVAcc::getLLK(const SegCluster &selectedSegments, const MixtureGD &model, FeatureBuffer &fs, double minLLK, double maxLLK)
{
    DeviceMixture dm(_srv, model);
    return accumulateStatLLK(fs, dm, selectedSegments, minLLK, maxLLK);
}
double TVAcc::verifyEMLK(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerFile, const std::vector<unsigned long> &rowOfFile,
                         unsigned long maxLLKcomputed, double minLLK, double maxLLK, std::vector<double> *perFile)
{
    if (rowOfFile.size() != segsPerFile.size()) throw Exception("verifyEMLK: one statistics row per file expected");
    const size_t n = std::min<size_t>(segsPerFile.size(), (size_t)maxLLKcomputed);
    std::vector<double> Sp;
    getMplusTW(Sp, std::vector<unsigned long>(rowOfFile.begin(), rowOfFile.begin() + n));   // every speaker model in one pass
    if (perFile) perFile->assign(n, 0.0);
    MixtureGD model = _ubm;                                       // loadMixtureGD(inputWorldFilename) per file in the reference
    DeviceMixture dm(_srv, model);
    double total = 0.0;
    for (size_t f = 0; f < n; ++f) {
        for (unsigned long c = 0; c < _n_distrib; ++c)
            for (unsigned long i = 0; i < _vectSize; ++i) model.setMean(c, Sp[f * _svSize + c * _vectSize + i], i);
        dm.update(model);
        const double llk = accumulateStatLLK(fs, dm, segsPerFile[f], minLLK, maxLLK);
        if (perFile) (*perFile)[f] = llk;
        total += llk;
    }
    return total;
}
void TVAcc::normTMatrix()
{
    _srv.check(gmmiv_tv_norm_t(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, _T.dev(), _ubm_invvar.cdev()));
}
void TVAcc::getWeightedCov(std::vector<double> &W, const std::vector<double> &weight)
{
    if (weight.size() != _n_distrib) throw Exception("getWeightedCov: one weight per distribution expected");
    W.assign(_rankT * _rankT, 0.0);
    _srv.check(gmmiv_tv_weighted_cov(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, _T.cdev(), weight.data(), W.data()));
}
void TVAcc::approximateTcTc(std::vector<double> &D, const std::vector<double> &Q)
{
    if (Q.size() != _rankT * _rankT) throw Exception("approximateTcTc: Q must be rankT x rankT");
    if (D.size() != _n_distrib * _rankT) D.assign(_n_distrib * _rankT, 0.0);
    _srv.check(gmmiv_tv_approximate_tctc(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankT, _T.cdev(), Q.data(), D.data()));
}
void TVAcc::estimateWUbmWeight(const std::vector<double> &W)
{
    if (W.size() != _rankT * _rankT) throw Exception("estimateWUbmWeight: W must be rankT x rankT");
    _W.assign(_W.size(), 0.0); // _W.setAllValues(0.0), :2353
    _srv.check(gmmiv_tv_estimate_w_ubm_weight(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankT, _statN.cdev(),
                                              _statF.cdev(), _T.cdev(), W.data(), _W.dev()));
}
void TVAcc::estimateWEigenDecomposition(const std::vector<double> &D, const std::vector<double> &Q)
{
    if (D.size() != _n_distrib * _rankT || Q.size() != _rankT * _rankT) throw Exception("estimateWEigenDecomposition: D is C x rankT, Q rankT x rankT");
    _srv.check(gmmiv_tv_estimate_w_eigen(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankT, _statN.cdev(),
                                         _statF.cdev(), _T.cdev(), D.data(), Q.data(), _W.dev()));
}


// ---- JFAAcc ----------------------------------------------------------------------------------------
JFAAcc::JFAAcc(GpuServer &srv, const MixtureGD &ubm, unsigned long rankEV, unsigned long rankEC,
               const std::vector<unsigned long> &sessionsPerSpeaker)
    : _srv(srv), _ubm(ubm), _dubm(srv, ubm), _rankEV(rankEV), _rankEC(rankEC), _n_speakers(sessionsPerSpeaker.size()), _n_sessions(0),
      _n_distrib(ubm.getDistribCount()), _vectSize(ubm.getVectSize()), _svSize(ubm.getDistribCount() * ubm.getVectSize())
{
    if (!rankEV || !rankEC) throw Exception("JFAAcc: eigenVoiceNumber and eigenChannelNumber must be positive");
    _sess_begin.assign(_n_speakers + 1, 0);
    for (unsigned long s = 0; s < _n_speakers; ++s) {
        if (!sessionsPerSpeaker[s]) throw Exception("JFAAcc: a speaker without session");
        _sess_begin[s + 1] = _sess_begin[s] + (int64_t)sessionsPerSpeaker[s];
        for (unsigned long k = 0; k < sessionsPerSpeaker[s]; ++k) _owner.push_back((int64_t)s);
    }
    _n_sessions = (unsigned long)_sess_begin[_n_speakers];
    for (DVec *v : {&_ubm_means, &_ubm_invvar, &_matN, &_N_h, &_F_X, &_F_X_h, &_cN, &_cN_h, &_cF_X, &_cF_X_h, &_V, &_matU, &_D, &_Y, &_matX, &_Z,
                    &_vEvT, &_uEuT, &_Aev, &_Cev, &_Aec, &_Cec, &_mdR, &_mdr, &_mdmw}) v->bind(srv);
    _ubm_means.set(_ubm.means());
    _ubm_invvar.set(_ubm.covInvs());
    _matN.assign(_n_speakers * _n_distrib, 0.0);
    _N_h.assign(_n_sessions * _n_distrib, 0.0);
    _F_X.assign(_n_speakers * _svSize, 0.0);
    _F_X_h.assign(_n_sessions * _svSize, 0.0);
    _V.assign(_rankEV * _svSize, 0.0);
    _matU.assign(_rankEC * _svSize, 0.0);
    _D.assign(_svSize, 0.0);
    _Y.assign(_n_speakers * _rankEV, 0.0);
    _matX.assign(_n_sessions * _rankEC, 0.0);
    _Z.assign(_n_speakers * _svSize, 0.0);
    _vEvT.assign(_n_distrib * gmmiv_tv_packed_len((int)_rankEV), 0.0);
    _uEuT.assign(_n_distrib * gmmiv_tv_packed_len((int)_rankEC), 0.0);
    resetTmpAcc();
}
void JFAAcc::resetTmpAcc()
{
    _Aev.assign(_n_distrib * gmmiv_tv_packed_len((int)_rankEV), 0.0);
    _Cev.assign(_rankEV * _svSize, 0.0);
    _Aec.assign(_n_distrib * gmmiv_tv_packed_len((int)_rankEC), 0.0);
    _Cec.assign(_rankEC * _svSize, 0.0);
}
void JFAAcc::computeAndAccumulateJFAStat(FeatureBuffer &fs, const std::vector<SegCluster> &segsPerSession)
{
    if (segsPerSession.size() != _n_sessions) throw Exception("computeAndAccumulateJFAStat: one SegCluster per session expected");
    SegCluster all;
    std::vector<int64_t> begin(_n_sessions + 1, 0);
    for (unsigned long h = 0; h < _n_sessions; ++h) {
        for (const Seg &s : segsPerSession[h]) all.push_back(s);
        begin[h + 1] = begin[h] + (int64_t)totalFrame(segsPerSession[h]);
    }
    unsigned long n = 0;
    const float *x = fs.select(all, n);
    if (&fs.server() != &_srv) fs.server().sync(); // the selection is enqueued on the feature buffer's stream
    // the frame loop (:544-575) runs once, per session, on the device; a speaker's rows are the sums of its sessions' rows
    FiniteScope fin(_srv, fs); // the frames may come from another server's buffer: the decision follows the buffer, not this context
    _srv.check(gmmiv_tv_stats(_srv.ctx(), _dubm.handle(), x, GMMIV_F32, (int64_t)n, (int64_t)fs.getVectSize(), begin.data(),
                              (int64_t)_n_sessions, _N_h.dev(), _F_X_h.dev()));
    // speaker rows = sums of the speaker's session rows (once per statistics pass; on the host views)
    const std::vector<double> &nh = _N_h.chost(), &fh = _F_X_h.chost();
    std::vector<double> nS(_n_speakers * _n_distrib, 0.0), fS(_n_speakers * _svSize, 0.0);
    for (unsigned long h = 0; h < _n_sessions; ++h) {
        const unsigned long s = (unsigned long)_owner[h];
        for (unsigned long k = 0; k < _n_distrib; ++k) nS[s * _n_distrib + k] += nh[h * _n_distrib + k];
        for (unsigned long k = 0; k < _svSize; ++k) fS[s * _svSize + k] += fh[h * _svSize + k];
    }
    _matN.set(nS); _F_X.set(fS);
}
void JFAAcc::setStats(const std::vector<double> &N, const std::vector<double> &N_h, const std::vector<double> &F_X,
                      const std::vector<double> &F_X_h)
{
    if (N.size() != _matN.size() || N_h.size() != _N_h.size() || F_X.size() != _F_X.size() || F_X_h.size() != _F_X_h.size())
        throw Exception("JFAAcc::setStats: dimension mismatch");
    _matN.set(N); _N_h.set(N_h); _F_X.set(F_X); _F_X_h.set(F_X_h);
}
void JFAAcc::storeAccs() { _cF_X.copyFrom(_F_X); _cF_X_h.copyFrom(_F_X_h); _cN_h.copyFrom(_N_h); _cN.copyFrom(_matN); }   // device-to-device
void JFAAcc::restoreAccs() { _matN.copyFrom(_cN); _N_h.copyFrom(_cN_h); _F_X.copyFrom(_cF_X); _F_X_h.copyFrom(_cF_X_h); }
void JFAAcc::loadEV(const std::vector<double> &V)
{
    if (V.size() != _V.size()) throw Exception("Incorrect dimension of EigenVoice Matrix");
    _V.set(V);
}
void JFAAcc::loadEC(const std::vector<double> &U)
{
    if (U.size() != _matU.size()) throw Exception("Incorrect dimension of EigenChannel Matrix");
    _matU.set(U);
}
void JFAAcc::loadD(const std::vector<double> &D)
{
    if (D.size() != _D.size()) throw Exception("Incorrect dimension of D Matrix");
    _D.set(D);
}
void JFAAcc::initD(double regulationFactor)
{
    const std::vector<double> &iv = _ubm_invvar.chost();
    std::vector<double> d(_svSize);
    for (unsigned long i = 0; i < _svSize; ++i) d[i] = sqrt(1.0 / (iv[i] * regulationFactor));
    _D.set(d);
}
void JFAAcc::estimateVEVT()
{
    _srv.check(gmmiv_tv_tett(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankEV, _V.cdev(), _ubm_invvar.cdev(), _vEvT.dev()));
}
void JFAAcc::estimateUEUT()
{
    _srv.check(gmmiv_tv_tett(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankEC, _matU.cdev(), _ubm_invvar.cdev(), _uEuT.dev()));
}
void JFAAcc::estimateYandV()
{
    _mdR.assign(_rankEV * _rankEV, 0.0); _mdr.assign(_rankEV, 0.0); _mdmw.assign(_rankEV, 0.0); // minimum-divergence sums: unused by JFA
    _srv.check(gmmiv_tv_estimate_a_and_c(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankEV, _matN.cdev(),
                                         _F_X.cdev(), _V.cdev(), _ubm_invvar.cdev(), _vEvT.cdev(), _Y.dev(), _Aev.dev(), _Cev.dev(),
                                         _mdR.dev(), _mdr.dev(), _mdmw.dev()));
}
void JFAAcc::estimateY()
{
    _srv.check(gmmiv_tv_estimate_w(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, (int)_rankEV, _matN.cdev(), _F_X.cdev(),
                                   _V.cdev(), _ubm_invvar.cdev(), _vEvT.cdev(), _Y.dev()));
}
void JFAAcc::estimateXandU()
{
    _mdR.assign(_rankEC * _rankEC, 0.0); _mdr.assign(_rankEC, 0.0); _mdmw.assign(_rankEC, 0.0);
    _srv.check(gmmiv_tv_estimate_a_and_c(_srv.ctx(), (int64_t)_n_sessions, (int)_n_distrib, (int)_vectSize, (int)_rankEC, _N_h.cdev(),
                                         _F_X_h.cdev(), _matU.cdev(), _ubm_invvar.cdev(), _uEuT.cdev(), _matX.dev(), _Aec.dev(),
                                         _Cec.dev(), _mdR.dev(), _mdr.dev(), _mdmw.dev()));
}
void JFAAcc::estimateX()
{
    _srv.check(gmmiv_tv_estimate_w(_srv.ctx(), (int64_t)_n_sessions, (int)_n_distrib, (int)_vectSize, (int)_rankEC, _N_h.cdev(), _F_X_h.cdev(),
                                   _matU.cdev(), _ubm_invvar.cdev(), _uEuT.cdev(), _matX.dev()));
}
void JFAAcc::estimateZandD()
{
    _srv.check(gmmiv_jfa_estimate_z_and_d(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.cdev(),
                                          _ubm_invvar.cdev(), _D.dev(), _Z.dev()));
}
void JFAAcc::estimateZ()
{
    _srv.check(gmmiv_jfa_estimate_z(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.cdev(),
                                    _ubm_invvar.cdev(), _D.cdev(), -1.0, _Z.dev()));
}
void JFAAcc::estimateZMAP(double tau)
{
    if (tau < 0.0) throw Exception("estimateZMAP: negative relevance factor");
    _srv.check(gmmiv_jfa_estimate_z(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.cdev(),
                                    _ubm_invvar.cdev(), _D.cdev(), tau, _Z.dev()));
}
void JFAAcc::updateVestimate()
{
    _srv.check(gmmiv_tv_update_t(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankEV, _Aev.cdev(), _Cev.cdev(), _V.dev()));
    _Cev.copyFrom(_V); // the reference leaves the new matrix in _Cev too (:3617-3618)
}
void JFAAcc::updateUestimate()
{
    _srv.check(gmmiv_tv_update_t(_srv.ctx(), (int)_n_distrib, (int)_vectSize, (int)_rankEC, _Aec.cdev(), _Cec.cdev(), _matU.dev()));
    _Cec.copyFrom(_matU);
}
void JFAAcc::substractMplusDZ()
{
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.dev(), nullptr,
                                  (int64_t)_n_speakers, _ubm_means.cdev(), 0, nullptr, nullptr, _D.cdev(), _Z.cdev()));
}
void JFAAcc::substractMplusVY()
{
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.dev(), nullptr,
                                  (int64_t)_n_speakers, _ubm_means.cdev(), (int)_rankEV, _V.cdev(), _Y.cdev(), nullptr, nullptr));
}
void JFAAcc::substractUX()
{
    _srv.check(gmmiv_jfa_subtract_sessions(_srv.ctx(), (int64_t)_n_speakers, _sess_begin.data(), (int)_n_distrib, (int)_vectSize, _N_h.cdev(),
                                           _F_X.dev(), (int)_rankEC, _matU.cdev(), _matX.cdev()));
}
void JFAAcc::substractMplusVYplusDZ()
{
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)_n_sessions, (int)_n_distrib, (int)_vectSize, _N_h.cdev(), _F_X_h.dev(), _owner.data(),
                                  (int64_t)_n_speakers, _ubm_means.cdev(), (int)_rankEV, _V.cdev(), _Y.cdev(), _D.cdev(), _Z.cdev()));
}
void JFAAcc::substractMplusUX()
{
    // The reference takes N_h (m + U x_h) of every session out of the SPEAKER statistics (:4344-4356).  sum_h N_h m = N m
    // (the speaker occupancies are the sums of their sessions'), the channel parts are substractUX.
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)_n_speakers, (int)_n_distrib, (int)_vectSize, _matN.cdev(), _F_X.dev(), nullptr,
                                  (int64_t)_n_speakers, _ubm_means.cdev(), 0, nullptr, nullptr, nullptr, nullptr));
    substractUX();
}
void JFAAcc::substractMplusDZByChannel()
{
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), (int64_t)_n_sessions, (int)_n_distrib, (int)_vectSize, _N_h.cdev(), _F_X_h.dev(), _owner.data(),
                                  (int64_t)_n_speakers, _ubm_means.cdev(), 0, nullptr, nullptr, _D.cdev(), _Z.cdev()));
}
void JFAAcc::orthonormalizeV()
{
    _srv.check(gmmiv_tv_orthonormalize_t(_srv.ctx(), (int)_rankEV, (int64_t)_svSize, _V.dev()));
}
void JFAAcc::getMplusVYplusDZ(std::vector<double> &Sp, unsigned long spk)
{
    if (spk >= _n_speakers) throw Exception("getMplusVYplusDZ: speaker index out of range");
    // Sp = 0 - (-1) (m + V y + D z): the subtraction with unit negative occupations builds the supervector
    std::vector<double> n1(_n_distrib, -1.0);
    Sp.assign(_svSize, 0.0);
    _srv.check(gmmiv_jfa_subtract(_srv.ctx(), 1, (int)_n_distrib, (int)_vectSize, n1.data(), Sp.data(), nullptr, 1, _ubm_means.cdev(), (int)_rankEV,
                                  _V.cdev(), _Y.cdev() + spk * _rankEV, _D.cdev(), _Z.cdev() + spk * _svSize));
}

std::vector<double> computeTestDotProduct(GpuServer &srv, JFAAcc &jfaAcc, const std::vector<double> &clientSV, unsigned long nClients)
{
    const unsigned long nTest = jfaAcc.getNSpeakers();
    if (jfaAcc.getNSessions() != nTest) throw Exception("computeTestDotProduct: one session per test segment expected");
    const unsigned long C = jfaAcc.getN().size() / nTest, SV = jfaAcc.getF_X().size() / nTest;
    if (clientSV.size() != nClients * SV) throw Exception("computeTestDotProduct: client supervector dimension mismatch");
    jfaAcc.estimateUEUT();                 // ComputeTest.cpp:319-331
    jfaAcc.estimateAndInverseL_EC();
    jfaAcc.substractMplusVYplusDZ();
    jfaAcc.estimateX();
    jfaAcc.substractMplusUX();
    // f_x /= sum_c N_c (:333-342), laid out [SV x nTest] for the product
    std::vector<double> ft(SV * nTest);
    for (unsigned long t = 0; t < nTest; ++t) {
        double sumN = 0.0;
        for (unsigned long i = 0; i < C; ++i) sumN += jfaAcc.getN()[t * C + i];
        for (unsigned long k = 0; k < SV; ++k) ft[k * nTest + t] = jfaAcc.getF_X()[t * SV + k] / sumN;
    }
    // scores = M F' (:352-357): clients x tests on the device, returned test-major
    std::vector<double> sc(nClients * nTest), out(nTest * nClients);
    srv.check(gmmiv_iv_normalize(srv.ctx(), (int)SV, (int)nClients, (int64_t)nTest, ft.data(), nullptr, clientSV.data(), 0, sc.data()));
    for (unsigned long c = 0; c < nClients; ++c)
        for (unsigned long t = 0; t < nTest; ++t) out[t * nClients + c] = sc[c * nTest + t];
    return out;
}

void eigenVoice(JFAAcc &jfaAcc, unsigned long nbIt, bool orthonormalizeV)
{
    jfaAcc.storeAccs();
    for (unsigned long it = 0; it < nbIt; ++it) {
        jfaAcc.estimateVEVT();
        jfaAcc.estimateAndInverseL_EV();
        jfaAcc.substractMplusDZ();
        jfaAcc.substractUX();
        jfaAcc.estimateYandV();
        jfaAcc.updateVestimate();
        if (orthonormalizeV) jfaAcc.orthonormalizeV();
        jfaAcc.resetTmpAcc();
        jfaAcc.restoreAccs();
    }
}
void eigenChannel(JFAAcc &jfaAcc, unsigned long nbIt)
{
    jfaAcc.storeAccs();                 // speaker factors first (EigenChannel.cpp:120-128)
    jfaAcc.estimateVEVT();
    jfaAcc.estimateAndInverseL_EV();
    jfaAcc.substractMplusDZ();
    jfaAcc.substractUX();
    jfaAcc.estimateY();
    jfaAcc.restoreAccs();
    jfaAcc.storeAccs();
    for (unsigned long it = 0; it < nbIt; ++it) {
        jfaAcc.estimateUEUT();
        jfaAcc.estimateAndInverseL_EC();
        jfaAcc.substractMplusVYplusDZ();
        jfaAcc.estimateXandU();
        jfaAcc.updateUestimate();
        jfaAcc.resetTmpAcc();
        jfaAcc.restoreAccs();
    }
}
void estimateDMatrix(JFAAcc &jfaAcc, unsigned long nbIt)
{
    jfaAcc.storeAccs();                 // y for every speaker (EstimateDMatrix.cpp:143-151)
    jfaAcc.estimateVEVT();
    jfaAcc.estimateAndInverseL_EV();
    jfaAcc.substractMplusDZ();
    jfaAcc.substractUX();
    jfaAcc.estimateY();
    jfaAcc.restoreAccs();
    jfaAcc.storeAccs();                 // x for every session (:163-170)
    jfaAcc.estimateUEUT();
    jfaAcc.estimateAndInverseL_EC();
    jfaAcc.substractMplusVYplusDZ();
    jfaAcc.estimateX();
    jfaAcc.restoreAccs();
    jfaAcc.storeAccs();
    for (unsigned long it = 0; it < nbIt; ++it) {   // :178-190
        jfaAcc.substractMplusVY();
        jfaAcc.substractUX();
        jfaAcc.estimateZandD();
        jfaAcc.resetTmpAcc();
        jfaAcc.restoreAccs();
    }
}
// ---- PldaDev ---------------------------------------------------------------------------------------
PldaDev::PldaDev(GpuServer &srv, unsigned long vectSize, const std::vector<double> &data, const std::vector<unsigned long> &sessionPerSpeaker)
    : _srv(srv), _vectSize(vectSize), _n_sessions(0), _data(srv), _session_per_speaker(sessionPerSpeaker)
{
    for (unsigned long v : sessionPerSpeaker) {
        if (v == 0) throw Exception("PldaDev: a speaker without session");
        _n_sessions += v;
    }
    if (_vectSize == 0 || data.size() != _vectSize * _n_sessions) throw Exception("PldaDev: data must be vectSize x n_sessions");
    _data.set(data);   // the development set lives on the device from here on
    computeAll();
}
std::vector<int64_t> PldaDev::sps64() const { return std::vector<int64_t>(_session_per_speaker.begin(), _session_per_speaker.end()); }
void PldaDev::computeAll()
{
    const std::vector<int64_t> sp = sps64();
    _mean.assign(_vectSize, 0.0);
    _speaker_means.assign(_vectSize * sp.size(), 0.0);
    _srv.check(gmmiv_dev_means(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), (int64_t)sp.size(), sp.data(), _mean.data(),
                               _speaker_means.data()));
}
void PldaDev::lengthNorm()
{
    DVec out(_srv);
    out.assign(_data.size(), 0.0);
    _srv.check(gmmiv_iv_normalize(_srv.ctx(), (int)_vectSize, (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), nullptr, nullptr, 1, out.dev()));
    _data.swap(out);
    computeAll();
}
void PldaDev::center(const std::vector<double> &mu)
{
    if (mu.size() != _vectSize) throw Exception("PldaDev::center: mean of the wrong size");
    DVec out(_srv);
    out.assign(_data.size(), 0.0);
    _srv.check(gmmiv_iv_normalize(_srv.ctx(), (int)_vectSize, (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), mu.data(), nullptr, 0, out.dev()));
    _data.swap(out);
    computeAll();
}
void PldaDev::centerPerSpeaker()
{
    const unsigned long k = _session_per_speaker.size();
    unsigned long s = 0;
    std::vector<double> &x = _data.host();
    for (unsigned long c = 0; c < k; ++c)
        for (unsigned long e = 0; e < _session_per_speaker[c]; ++e, ++s)
            for (unsigned long d = 0; d < _vectSize; ++d) x[d * _n_sessions + s] -= _speaker_means[d * k + c];
}
void PldaDev::rotateLeft(const std::vector<double> &M, unsigned long rows)
{
    if (M.size() != rows * _vectSize) throw Exception("Rotation dimension mismatch !");
    DVec out(_srv);
    out.assign(rows * _n_sessions, 0.0);
    _srv.check(gmmiv_iv_normalize(_srv.ctx(), (int)_vectSize, (int)rows, (int64_t)_n_sessions, _data.cdev(), nullptr, M.data(), 0, out.dev()));
    _data.swap(out);
    _vectSize = rows;
    computeAll();
}
void PldaDev::computeCovMat(std::vector<double> &Sigma, std::vector<double> &W, std::vector<double> &B)
{
    const std::vector<int64_t> sp = sps64();
    const size_t dd = _vectSize * _vectSize;
    Sigma.assign(dd, 0.0); W.assign(dd, 0.0); B.assign(dd, 0.0);
    _srv.check(gmmiv_dev_cov_mat(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), (int64_t)sp.size(), sp.data(), Sigma.data(),
                                 W.data(), B.data()));
}
void PldaDev::computeWccnChol(std::vector<double> &WCCN)
{
    const std::vector<int64_t> sp = sps64();
    WCCN.assign(_vectSize * _vectSize, 0.0);
    _srv.check(gmmiv_dev_wccn_chol(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), (int64_t)sp.size(), sp.data(), WCCN.data()));
}
void PldaDev::computeMahalanobis(std::vector<double> &M)
{
    const std::vector<int64_t> sp = sps64();
    M.assign(_vectSize * _vectSize, 0.0);
    _srv.check(gmmiv_dev_mahalanobis(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), (int64_t)sp.size(), sp.data(), M.data()));
}
void PldaDev::computeScatterMat(std::vector<double> &SB, std::vector<double> &SW)
{
    const std::vector<int64_t> sp = sps64();
    SB.assign(_vectSize * _vectSize, 0.0); SW.assign(_vectSize * _vectSize, 0.0);
    _srv.check(gmmiv_dev_scatter_mat(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.cdev(), (int64_t)sp.size(), sp.data(), SB.data(),
                                     SW.data()));
}
void PldaDev::computeLDA(std::vector<double> &ldaMat, unsigned long ldaRank, bool scatterMatrices)
{
    std::vector<double> Sigma, W, B;
    if (scatterMatrices) computeScatterMat(B, W); // ldaMode scatterMatrices, :1389-1392
    else computeCovMat(Sigma, W, B);
    ldaMat.assign(ldaRank * _vectSize, 0.0);
    _srv.check(gmmiv_dev_lda(_srv.ctx(), (int)_vectSize, W.data(), B.data(), (int)ldaRank, ldaMat.data(), nullptr));
}
void PldaDev::sphericalNuisanceNormalization(unsigned long nbIt, bool sphNorm, std::vector<std::vector<double> > &mats,
                                             std::vector<std::vector<double> > &means)
{
    mats.clear(); means.clear();
    for (unsigned long it = 0; it < nbIt; ++it) {
        std::vector<double> Sigma, W, B, M(_vectSize * _vectSize);
        computeCovMat(Sigma, W, B);
        _srv.check(gmmiv_dev_efr_matrix(_srv.ctx(), (int)_vectSize, sphNorm ? W.data() : Sigma.data(), M.data()));
        mats.push_back(M);
        means.push_back(_mean);
        const std::vector<double> mu(_mean); // center() recomputes _mean
        center(mu);
        rotateLeft(M, _vectSize);
        lengthNorm();
    }
}
void PldaDev::emIteration(unsigned long rankF, unsigned long rankG, std::vector<double> &F, std::vector<double> &G, std::vector<double> &Sigma,
                          std::vector<double> &Delta, const std::vector<int64_t> &sp)
{
    _srv.check(gmmiv_plda_em_iteration(_srv.ctx(), (int)_vectSize, (int64_t)_n_sessions, _data.dev(), (int64_t)sp.size(), sp.data(), (int)rankF,
                                       (int)rankG, F.data(), G.data(), Sigma.data(), Delta.data()));
    computeAll(); // _Dev.center(_Delta) ends with computeAll(), :466-474
}
void PldaDev::applySphericalNuisanceNormalization(const std::vector<std::vector<double> > &mats, const std::vector<std::vector<double> > &means)
{
    if (mats.size() != means.size()) throw Exception("applySphericalNuisanceNormalization: one mean per matrix expected");
    for (size_t it = 0; it < mats.size(); ++it) {
        center(means[it]);
        rotateLeft(mats[it], mats[it].size() / _vectSize);
        lengthNorm();
    }
}

// ---- PldaModel (training) --------------------------------------------------------------------------
PldaModel::PldaModel(PldaDev &dev, unsigned long rankF, unsigned long rankG, const std::vector<double> &F, const std::vector<double> &G,
                     const std::vector<double> &Sigma)
    : _Dev(dev), _rankF(rankF), _rankG(rankG), _vectSize(dev.getVectSize()), _F(F), _G(G), _Sigma(Sigma), _Delta(dev.getVectSize(), 0.0),
      _originalMean(dev.getMean())
{
    if (_F.size() != _vectSize * _rankF || _G.size() != _vectSize * _rankG || _Sigma.size() != _vectSize * _vectSize)
        throw Exception("PldaModel: F is vectSize x rankF, G vectSize x rankG, Sigma vectSize x vectSize");
}
void PldaModel::em_iteration()
{
    std::vector<int64_t> sp(_Dev.getSpeakerNumber());
    for (unsigned long i = 0; i < sp.size(); ++i) sp[i] = (int64_t)_Dev.getSpeakerSessionNumber(i);
    // gmmiv_plda_em_iteration centres the data by Delta in place (_Dev.center(_Delta)); the reference recomputes the
    // means of _Dev there, which nothing in the iteration reads
    _Dev.emIteration(_rankF, _rankG, _F, _G, _Sigma, _Delta, sp);
}

// ---- IvTest ----------------------------------------------------------------------------------------
namespace {
void twoCovModel(GpuServer &srv, unsigned long dim, const std::vector<double> &W, const std::vector<double> &B, std::vector<double> &G,
                 std::vector<double> &H)
{
    srv.check(gmmiv_twocov_model(srv.ctx(), (int)dim, W.data(), B.data(), G.data(), H.data()));
}
// center (optional) + rotateLeft (optional) + lengthNorm (optional) of the columns of X [dim x n] -> [rows x n]
void normaliseColumns(GpuServer &srv, std::vector<double> &X, unsigned long &dim, unsigned long n, const std::vector<double> *mean,
                      const std::vector<double> *M, unsigned long rows, bool lengthNorm)
{
    const unsigned long dout = M ? rows : dim;
    std::vector<double> out(dout * n);
    srv.check(gmmiv_iv_normalize(srv.ctx(), (int)dim, (int)dout, (int64_t)n, X.data(), mean ? mean->data() : nullptr, M ? M->data() : nullptr,
                                 lengthNorm ? 1 : 0, out.data()));
    X.swap(out);
    dim = dout;
}
} // namespace

std::vector<double> ivTest(GpuServer &srv, const IvTestCfg &cfg, PldaDev &dev, std::vector<double> enrol,
                           const std::vector<unsigned long> &enrolPerModel, std::vector<double> test, unsigned long nTest,
                           const std::vector<double> &pldaF, const std::vector<double> &pldaG, const std::vector<double> &pldaSigma)
{
    unsigned long dimE = dev.getVectSize(), dimT = dev.getVectSize(), nEnrol = 0;
    for (unsigned long v : enrolPerModel) nEnrol += v;
    const unsigned long nModels = enrolPerModel.size();
    if (enrol.size() != dimE * nEnrol || test.size() != dimT * nTest) throw Exception("ivTest: enrol / test must be vectSize x count");
    // 1. normalisation parameters on the development set (IvTest.cpp:131-160, 262-290)
    std::vector<std::vector<double> > mats, means;
    std::vector<double> ldaMat;
    if (cfg.ivNorm) {
        if (cfg.ivNormIterationNb > 0) dev.sphericalNuisanceNormalization(cfg.ivNormIterationNb, cfg.sphNorm, mats, means);
        if (cfg.LDA) {
            dev.computeLDA(ldaMat, cfg.ldaRank);
            dev.rotateLeft(ldaMat, cfg.ldaRank);
        }
    }
    // 2. back-end matrices on the normalised development set (:200-258) or the PLDA model (:262-310)
    std::vector<double> wccn, mah, W, B, Sigma, F, G, Sg;
    if (cfg.scoring == "cosine") { if (cfg.WCCN) dev.computeWccnChol(wccn); }
    else if (cfg.scoring == "mahalanobis") dev.computeMahalanobis(mah);
    else if (cfg.scoring == "2cov") dev.computeCovMat(Sigma, W, B);
    else if (cfg.scoring == "plda") {
        // plda.updateModel + centerData (:292-294): the data is centred on its own mean before the EM
        const std::vector<double> mu(dev.getMean());
        dev.center(mu);
        PldaModel plda(dev, cfg.pldaRankF, cfg.pldaRankG, pldaF, pldaG, pldaSigma);
        for (unsigned long it = 0; it < cfg.pldaNbIt; ++it) plda.em_iteration();
        F = plda.getF(); G = plda.getG(); Sg = plda.getSigma();
    } else throw Exception("Scoring option is invalid, must be: cosine OR mahalanobis OR 2cov OR plda");
    // 3. the same normalisation on the enrolment and test vectors (:312-328)
    if (cfg.ivNorm) {
        for (size_t it = 0; it < mats.size(); ++it) {
            normaliseColumns(srv, enrol, dimE, nEnrol, &means[it], &mats[it], mats[it].size() / dimE, true);
            normaliseColumns(srv, test, dimT, nTest, &means[it], &mats[it], mats[it].size() / dimT, true);
        }
        if (cfg.LDA) {
            normaliseColumns(srv, enrol, dimE, nEnrol, nullptr, &ldaMat, cfg.ldaRank, false);
            normaliseColumns(srv, test, dimT, nTest, nullptr, &ldaMat, cfg.ldaRank, false);
        }
    }
    if (cfg.scoring == "cosine" && cfg.WCCN) {
        normaliseColumns(srv, enrol, dimE, nEnrol, nullptr, &wccn, dimE, false);
        normaliseColumns(srv, test, dimT, nTest, nullptr, &wccn, dimT, false);
    }
    std::vector<double> FTJ, FTJF;
    if (cfg.scoring == "plda") { // pldaNativeScoring: rotateLeft(FTJ), PldaTools.cpp:4494-4504
        const unsigned long rf = cfg.pldaRankF, rg = cfg.pldaRankG;
        FTJ.assign(rf * dimE, 0.0); FTJF.assign(rf * rf, 0.0);
        srv.check(gmmiv_plda_precompute(srv.ctx(), (int)dimE, (int)rf, (int)rg, F.data(), rg ? G.data() : nullptr, Sg.data(), FTJ.data(), FTJF.data()));
        normaliseColumns(srv, enrol, dimE, nEnrol, nullptr, &FTJ, rf, false);
        normaliseColumns(srv, test, dimT, nTest, nullptr, &FTJ, rf, false);
    }
    // 4. one vector per model: mean of its enrolment vectors (sum + count for plda)
    std::vector<double> models(dimE * nModels, 0.0);
    std::vector<int64_t> nsess(nModels);
    unsigned long s0 = 0;
    for (unsigned long m = 0; m < nModels; ++m) {
        const unsigned long ns = enrolPerModel[m];
        if (ns == 0) throw Exception("ivTest: a model without enrolment vector");
        nsess[m] = (int64_t)ns;
        for (unsigned long d = 0; d < dimE; ++d) {
            double a = 0.0;
            for (unsigned long e = 0; e < ns; ++e) a += enrol[d * nEnrol + s0 + e];
            models[d * nModels + m] = cfg.scoring == "plda" ? a : a / (double)ns;
        }
        s0 += ns;
    }
    // 5. scoring (:330-395)
    std::vector<double> scores(nModels * nTest);
    if (cfg.scoring == "cosine")
        srv.check(gmmiv_score_cosine(srv.ctx(), (int)dimE, (int64_t)nModels, (int64_t)nTest, models.data(), test.data(), scores.data()));
    else if (cfg.scoring == "mahalanobis")
        srv.check(gmmiv_score_mahalanobis(srv.ctx(), (int)dimE, (int64_t)nModels, (int64_t)nTest, models.data(), test.data(), mah.data(), scores.data()));
    else if (cfg.scoring == "2cov") {
        // G and H of twoCovScoring (:4089-4125) from W and B
        const size_t dd = dimE * dimE;
        std::vector<double> Gm(dd), Hm(dd);
        twoCovModel(srv, dimE, W, B, Gm, Hm);
        srv.check(gmmiv_score_twocov(srv.ctx(), (int)dimE, (int64_t)nModels, (int64_t)nTest, models.data(), test.data(), Gm.data(), Hm.data(), scores.data()));
    } else
        srv.check(gmmiv_score_plda(srv.ctx(), (int)cfg.pldaRankF, (int64_t)nModels, (int64_t)nTest, models.data(), nsess.data(), test.data(), FTJF.data(), scores.data()));
    return scores;
}

void computeEigenProblem(const std::vector<double> &EP, unsigned long n, std::vector<double> &eigenVect, std::vector<double> &eigenVal,
                         unsigned long rank)
{
    if (EP.size() != n * n || rank > n) throw Exception("computeEigenProblem: EP must be n x n and rank <= n");
    std::vector<double> a(EP), v(n * n, 0.0);
    for (unsigned long i = 0; i < n; ++i) v[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (unsigned long i = 0; i < n; ++i) {
            diag += a[i * n + i] * a[i * n + i];
            for (unsigned long j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
        }
        if (off <= 1e-30 * (diag + off)) break;
        for (unsigned long p = 0; p + 1 < n; ++p)
            for (unsigned long q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (unsigned long k = 0; k < n; ++k) { // columns p, q
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = cs * akp - sn * akq;
                    a[k * n + q] = sn * akp + cs * akq;
                }
                for (unsigned long k = 0; k < n; ++k) { // rows p, q
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = cs * apk - sn * aqk;
                    a[q * n + k] = sn * apk + cs * aqk;
                }
                for (unsigned long k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = cs * vkp - sn * vkq;
                    v[k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
    std::vector<unsigned long> order(n);
    for (unsigned long i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](unsigned long x, unsigned long y) { return a[x * n + x] > a[y * n + y]; });
    eigenVect.assign(n * rank, 0.0);
    eigenVal.assign(rank, 0.0);
    for (unsigned long j = 0; j < rank; ++j) {
        eigenVal[j] = a[order[j] * n + order[j]];
        for (unsigned long k = 0; k < n; ++k) eigenVect[k * rank + j] = v[k * n + order[j]];
    }
}

} // namespace liagpu
