// capi_host.cpp -- flat C entry points over liatools_gpu (used by the Python tests / tools; a C++
// caller includes liatools_gpu.h directly).  Every function returns 0 or -1 (+ liagpu_last_error()),
// mirroring the reference tools' "catch, print, carry on" drivers.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>

#include "liatools_gpu.h"
#include "io.h"

using namespace liagpu;

static thread_local std::string g_err;
#define GUARD(...)                                          \
    try { __VA_ARGS__; return 0; }                              \
    catch (const std::exception &e) { g_err = e.what(); return -1; }

static SegCluster make_cluster(const long *begin, const long *len, long n)
{
    SegCluster c(n);
    for (long i = 0; i < n; ++i) { c[i].begin = (unsigned long)begin[i]; c[i].length = (unsigned long)len[i]; c[i].source = 0; }
    return c;
}
static MixtureGD make_mixture(int C, int D, const double *w, const double *mean, const double *cov)
{
    MixtureGD m(C, D);
    m.weights().assign(w, w + C);
    m.means().assign(mean, mean + (size_t)C * D);
    m.covs().assign(cov, cov + (size_t)C * D);
    m.computeAll();
    return m;
}

extern "C" {

const char *liagpu_last_error(void) { return g_err.c_str(); }
int liagpu_train_world_timed(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                             int C, double *w, double *mean, double *cov, int nbTrainIt, double baggedFrameProbability,
                             double initVarFloor, double finalVarFloor, double initVarCeil, double finalVarCeil,
                             long initRand, double *global_mean_out, double *global_cov_out, double *llk_it_out, double *it_ms_out);
int liagpu_compute_test_timed(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                              int C, const double *w_world, const double *mean_world, const double *cov_world, int nClients,
                              const double *w_cl, const double *mean_cl, const double *cov_cl, int topDistribsCount,
                              int complete, double minLLK, double maxLLK, int segmentalMode, double *llr_out, int reps, double *ms_out);
int liagpu_iv_extract_timed(int device, const float *x, long T, int D, const long *utt_begin, long U, int C, const double *w,
                            const double *mean, const double *cov, int R, const double *Tmat, double *W_out, double *N_out,
                            double *F_out, int reps, double *ms_out);

// TrainWorld: computeMeanCov -> trainModelStream (TrainWorld.cpp:101-191 minus file I/O and mixtureInit)
int liagpu_train_world(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                       int C, double *w, double *mean, double *cov, int nbTrainIt, double baggedFrameProbability,
                       double initVarFloor, double finalVarFloor, double initVarCeil, double finalVarCeil,
                       long initRand, double *global_mean_out, double *global_cov_out, double *llk_it_out)
{
    return liagpu_train_world_timed(device, x, T, D, seg_begin, seg_len, nseg, C, w, mean, cov, nbTrainIt, baggedFrameProbability, initVarFloor,
                                    finalVarFloor, initVarCeil, finalVarCeil, initRand, global_mean_out, global_cov_out, llk_it_out, nullptr);
}

// the same, with the wall time of every iteration in it_ms_out[nbTrainIt] (nullable): what bench.py's host_layer block reports
int liagpu_train_world_timed(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                             int C, double *w, double *mean, double *cov, int nbTrainIt, double baggedFrameProbability,
                             double initVarFloor, double finalVarFloor, double initVarCeil, double finalVarCeil,
                             long initRand, double *global_mean_out, double *global_cov_out, double *llk_it_out, double *it_ms_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        std::vector<double> gm, gc;
        std::vector<double> itMs;
        computeMeanCov(fs, segs, gm, gc);
        if (global_mean_out) memcpy(global_mean_out, gm.data(), D * sizeof(double));
        if (global_cov_out) memcpy(global_cov_out, gc.data(), D * sizeof(double));
        MixtureGD world = make_mixture(C, D, w, mean, cov);
        TrainCfg cfg;
        cfg.nbTrainIt = nbTrainIt; cfg.baggedFrameProbability = baggedFrameProbability;
        cfg.initVarianceFlooring = initVarFloor; cfg.finalVarianceFlooring = finalVarFloor;
        cfg.initVarianceCeiling = initVarCeil; cfg.finalVarianceCeiling = finalVarCeil;
        cfg.initRand = (unsigned long)initRand;
        if (it_ms_out) cfg.iterationMs = &itMs;
        std::vector<double> llk = trainModelStream(cfg, fs, segs, gc, world);
        memcpy(w, world.weights().data(), C * sizeof(double));
        memcpy(mean, world.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, world.covs().data(), (size_t)C * D * sizeof(double));
        if (llk_it_out) memcpy(llk_it_out, llk.data(), llk.size() * sizeof(double));
        if (it_ms_out) memcpy(it_ms_out, itMs.data(), itMs.size() * sizeof(double));
    })
}

// TrainWorld (TrainWorld.cpp:101-191) with NO initial model ("World model init from scratch", :175-178): global mean / covariance
// (or use01: 0 / 1, :158-169), mixtureInit with C components, trainModelStream.  w / mean / cov [C], [C x D], [C x D] are outputs only.
int liagpu_train_world_scratch(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                               int C, double nbFrameToSelect, int use01, double *w, double *mean, double *cov, int nbTrainIt,
                               double baggedFrameProbability, double initVarFloor, double finalVarFloor, double initVarCeil,
                               double finalVarCeil, long initRand, double *global_cov_out, double *llk_it_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        std::vector<double> gm(D, 0.0), gc(D, 1.0);                    // initialize01
        if (!use01) computeMeanCov(fs, segs, gm, gc);
        if (global_cov_out) memcpy(global_cov_out, gc.data(), D * sizeof(double));
        MixtureGD world((unsigned long)C, (unsigned long)D);
        MixtureInitCfg icfg;
        icfg.nbFrameToSelect = nbFrameToSelect;
        mixtureInit(fs, segs, 1.0, world, gc, icfg);
        TrainCfg cfg;
        cfg.nbTrainIt = nbTrainIt; cfg.baggedFrameProbability = baggedFrameProbability;
        cfg.initVarianceFlooring = initVarFloor; cfg.finalVarianceFlooring = finalVarFloor;
        cfg.initVarianceCeiling = initVarCeil; cfg.finalVarianceCeiling = finalVarCeil;
        cfg.initRand = (unsigned long)initRand;
        std::vector<double> llk = trainModelStream(cfg, fs, segs, gc, world);
        memcpy(w, world.weights().data(), C * sizeof(double));
        memcpy(mean, world.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, world.covs().data(), (size_t)C * D * sizeof(double));
        if (llk_it_out) memcpy(llk_it_out, llk.data(), llk.size() * sizeof(double));
    })
}

// TrainWorld over nStream input streams ("inputStreamList" / "weightStreamList", TrainWorld.cpp:123-137): computeMeanCov over all
// streams, trainModelStream(fsTab, segTab, weightTab) with componentReduction / normalizeModel.  weight == NULL: 1 / nStream.
// opts[5] = {componentReduction, targetMixtureDistribCount, normalizeModel, normalizeModelMeanOnly, normalizeModelNbIt}.
// w / mean / cov hold the C initial components on entry and the *C_out final ones (<= C) on return.
int liagpu_train_world_streams(int device, int nStream, const float *const *x, const long *T, int D, const long *const *seg_begin,
                               const long *const *seg_len, const long *nseg, const double *weight, int C, double *w, double *mean,
                               double *cov, int nbTrainIt, double baggedFrameProbability, double initVarFloor, double finalVarFloor,
                               double initVarCeil, double finalVarCeil, long initRand, const long *opts, long *C_out,
                               double *global_mean_out, double *global_cov_out, double *llk_it_out)
{
    GUARD({
        if (nStream <= 0) throw Exception("TrainWorld error:no input stream");
        GpuServer srv(device);
        std::vector<std::unique_ptr<FeatureBuffer> > fsTab;
        std::vector<SegCluster> segTab(nStream);
        std::vector<TrainStream> streams(nStream);
        for (int i = 0; i < nStream; ++i) {
            fsTab.emplace_back(new FeatureBuffer(srv, x[i], (unsigned long)T[i], (unsigned long)D));
            segTab[i] = make_cluster(seg_begin[i], seg_len[i], nseg[i]);
        }
        for (int i = 0; i < nStream; ++i) {
            streams[i].fs = fsTab[i].get(); streams[i].segs = &segTab[i];
            streams[i].weight = weight ? weight[i] : 1.0 / (double)nStream;
        }
        std::vector<double> gm, gc;
        computeMeanCov(streams, gm, gc);
        if (global_mean_out) memcpy(global_mean_out, gm.data(), D * sizeof(double));
        if (global_cov_out) memcpy(global_cov_out, gc.data(), D * sizeof(double));
        MixtureGD world = make_mixture(C, D, w, mean, cov);
        TrainCfg cfg;
        cfg.nbTrainIt = nbTrainIt; cfg.baggedFrameProbability = baggedFrameProbability;
        cfg.initVarianceFlooring = initVarFloor; cfg.finalVarianceFlooring = finalVarFloor;
        cfg.initVarianceCeiling = initVarCeil; cfg.finalVarianceCeiling = finalVarCeil;
        cfg.initRand = (unsigned long)initRand;
        if (opts) {
            cfg.componentReduction = opts[0] != 0; cfg.targetDistribCount = (unsigned long)opts[1];
            cfg.normalizeModel = opts[2] != 0; cfg.normalizeModelMeanOnly = opts[3] != 0; cfg.normalizeModelNbIt = (unsigned long)opts[4];
        }
        std::vector<double> llk = trainModelStream(cfg, streams, gc, world);
        const unsigned long Co = world.getDistribCount();
        if (C_out) *C_out = (long)Co;
        memcpy(w, world.weights().data(), Co * sizeof(double));
        memcpy(mean, world.means().data(), (size_t)Co * D * sizeof(double));
        memcpy(cov, world.covs().data(), (size_t)Co * D * sizeof(double));
        if (llk_it_out) memcpy(llk_it_out, llk.data(), llk.size() * sizeof(double));
    })
}

// mixtureInit over nStream input streams (TrainTools.cpp:674-766, the call of TrainWorld.cpp:177): w / mean / cov [C], [C x D], [C x D] out;
// counts [C] (nullable) = frames picked per component over all streams.  weight == NULL: 1 / nStream.
int liagpu_mixture_init_streams(int device, int nStream, const float *const *x, const long *T, int D, const long *const *seg_begin,
                                const long *const *seg_len, const long *nseg, const double *weight, int C, const double *global_cov,
                                double nbFrameToSelect, long minLen, long maxLen, double *w, double *mean, double *cov, long *counts)
{
    GUARD({
        if (nStream <= 0) throw Exception("mixtureInit: no input stream");
        GpuServer srv(device);
        std::vector<std::unique_ptr<FeatureBuffer> > fsTab;
        std::vector<SegCluster> segTab(nStream);
        std::vector<TrainStream> streams(nStream);
        for (int i = 0; i < nStream; ++i) {
            fsTab.emplace_back(new FeatureBuffer(srv, x[i], (unsigned long)T[i], (unsigned long)D));
            segTab[i] = make_cluster(seg_begin[i], seg_len[i], nseg[i]);
        }
        for (int i = 0; i < nStream; ++i) {
            streams[i].fs = fsTab[i].get(); streams[i].segs = &segTab[i];
            streams[i].weight = weight ? weight[i] : 1.0 / (double)nStream;
        }
        MixtureGD world((unsigned long)C, (unsigned long)D);
        MixtureInitCfg cfg;
        cfg.nbFrameToSelect = nbFrameToSelect; cfg.baggedMinimalLength = (unsigned long)minLen; cfg.baggedMaximalLength = (unsigned long)maxLen;
        std::vector<unsigned long> cnt;
        mixtureInit(streams, world, std::vector<double>(global_cov, global_cov + D), cfg, &cnt);
        memcpy(w, world.weights().data(), C * sizeof(double));
        memcpy(mean, world.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, world.covs().data(), (size_t)C * D * sizeof(double));
        if (counts) for (int c = 0; c < C; ++c) counts[c] = (long)cnt[c];
    })
}

// meanLikelihood over several feature servers (GeneralTools.cpp:599-607), optionally weighted by one decision value per server (:610-624)
int liagpu_mean_llk_streams(int device, int nStream, const float *const *x, const long *T, int D, const long *const *seg_begin,
                            const long *const *seg_len, const long *nseg, const double *decision, int C, const double *w, const double *mean,
                            const double *cov, double minLLK, double maxLLK, double *out)
{
    GUARD({
        GpuServer srv(device);
        std::vector<std::unique_ptr<FeatureBuffer> > fsTab;
        std::vector<SegCluster> segTab(nStream);
        std::vector<TrainStream> streams(nStream);
        for (int i = 0; i < nStream; ++i) {
            fsTab.emplace_back(new FeatureBuffer(srv, x[i], (unsigned long)T[i], (unsigned long)D));
            segTab[i] = make_cluster(seg_begin[i], seg_len[i], nseg[i]);
        }
        for (int i = 0; i < nStream; ++i) { streams[i].fs = fsTab[i].get(); streams[i].segs = &segTab[i]; }
        MixtureGD model = make_mixture(C, D, w, mean, cov);
        DeviceMixture dm(srv, model);
        *out = decision ? meanLikelihood(streams, dm, std::vector<double>(decision, decision + nStream), minLLK, maxLLK)
                        : meanLikelihood(streams, dm, minLLK, maxLLK);
    })
}

// selectComponent(nbTop) + reduceModel + normalizeWeights, then (optionally) normalizeMixture to N(0, 1): the model edits of
// TrainTools.cpp:1078-1098 on their own (host arithmetic only -- no device is touched)
int liagpu_model_reduce_normalize(int C, int D, double *w, double *mean, double *cov, long nbTop, int normalize, int meanOnly, long nbIt,
                                  long *order_out)
{
    GUARD({
        MixtureGD m = make_mixture(C, D, w, mean, cov);
        if (order_out) {
            const std::vector<unsigned long> o = sortByWeight(m);
            for (int i = 0; i < C; ++i) order_out[i] = (long)o[i];
        }
        if (nbTop > 0 && nbTop < C) {
            std::vector<bool> sel;
            const unsigned long n = selectComponent(sel, (unsigned long)nbTop, m);
            MixtureGD out(n, (unsigned long)D);
            (void)reduceModel(sel, m, out);
            normalizeWeights(out);
            m = out;
        }
        if (normalize) normalizeMixture(m, std::vector<double>(), std::vector<double>(), true, (unsigned long)nbIt, meanOnly != 0);
        const unsigned long Co = m.getDistribCount();
        memcpy(w, m.weights().data(), Co * sizeof(double));
        memcpy(mean, m.means().data(), (size_t)Co * D * sizeof(double));
        memcpy(cov, m.covs().data(), (size_t)Co * D * sizeof(double));
    })
}

// mixtureInit (TrainTools.cpp:674-766 multi-stream form with one stream when single_stream == 0, :619-672 when 1): the
// start-from-scratch model of TrainWorld.  param = nbFrameToSelect (multi) or baggedFrameProbabilityInit (single).
// w / mean / cov [C], [C x D], [C x D] out; counts [C] (nullable) = frames picked per component.
int liagpu_mixture_init(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg, int C,
                        int single_stream, double param, double stream_weight, long min_len, long max_len, const double *global_cov,
                        double *w, double *mean, double *cov, long *counts)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD world((unsigned long)C, (unsigned long)D);
        MixtureInitCfg cfg;
        cfg.baggedMinimalLength = (unsigned long)min_len; cfg.baggedMaximalLength = (unsigned long)max_len;
        std::vector<unsigned long> cnt;
        const std::vector<double> gc(global_cov, global_cov + D);
        if (single_stream) { cfg.baggedFrameProbabilityInit = param; mixtureInitSingleStream(fs, world, segs, gc, cfg, &cnt); }
        else { cfg.nbFrameToSelect = param; mixtureInit(fs, segs, stream_weight, world, gc, cfg, &cnt); }
        memcpy(w, world.weights().data(), C * sizeof(double));
        memcpy(mean, world.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, world.covs().data(), (size_t)C * D * sizeof(double));
        if (counts) for (int k = 0; k < C; ++k) counts[k] = (long)cnt[k];
    })
}

// TVAcc::verifyEMLK (AccumulateTVStat.cpp:1654-1688) on given i-vectors: file f = frames [file_begin[f], file_begin[f+1]) of x,
// its speaker model = UBM with means m + T^T W[row_of_file[f]]; llk_out [nfiles] = getLLK per file, *total = their sum.
// supervectors_out (nullable) [nfiles x C*D] = getMplusTW of each file's row.
int liagpu_tv_verify_emlk(int device, const float *x, long T, int D, const long *file_begin, long nfiles, const long *row_of_file, int C,
                          const double *w, const double *mean, const double *cov, int R, const double *Tmat, long U, const double *W,
                          long max_llk_computed, double minLLK, double maxLLK, double *llk_out, double *total, double *supervectors_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        TVAcc tv(srv, make_mixture(C, D, w, mean, cov), (unsigned long)R, (unsigned long)U);
        tv.loadT(std::vector<double>(Tmat, Tmat + (size_t)R * C * D));
        memcpy(tv.getW().data(), W, (size_t)U * R * sizeof(double));
        std::vector<SegCluster> segs(nfiles);
        std::vector<unsigned long> rows(nfiles);
        for (long f = 0; f < nfiles; ++f) {
            Seg s; s.begin = (unsigned long)file_begin[f]; s.length = (unsigned long)(file_begin[f + 1] - file_begin[f]);
            segs[f].push_back(s);
            rows[f] = (unsigned long)row_of_file[f];
        }
        std::vector<double> per;
        *total = tv.verifyEMLK(fs, segs, rows, (unsigned long)max_llk_computed, minLLK, maxLLK, &per);
        for (size_t f = 0; f < per.size(); ++f) llk_out[f] = per[f];
        if (supervectors_out) {
            std::vector<double> Sp;
            tv.getMplusTW(Sp, rows);
            memcpy(supervectors_out, Sp.data(), Sp.size() * sizeof(double));
        }
    })
}

// accumulateStatLLK over a cluster (AccumulateStat.cpp:69-94): getMeanLLK of the selected frames, clamped per frame
int liagpu_mean_llk(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg, int C,
                    const double *w, const double *mean, const double *cov, double minLLK, double maxLLK, double *out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        DeviceMixture dm(srv, make_mixture(C, D, w, mean, cov));
        *out = accumulateStatLLK(fs, dm, make_cluster(seg_begin, seg_len, nseg), minLLK, maxLLK);
    })
}

// TrainTarget (TrainTarget.cpp:73-278 minus file I/O): client = MAP-adapt(world) on the selected frames
int liagpu_train_target(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                        int C, const double *w, const double *mean, const double *cov, int nbTrainIt, double meanReg,
                        double *w_out, double *mean_out, double *cov_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD world = make_mixture(C, D, w, mean, cov);
        MixtureGD client = world;
        MAPCfg cfg;
        cfg.nbTrainIt = nbTrainIt; cfg.meanReg = meanReg;
        adaptModel(fs, segs, world, client, cfg);
        memcpy(w_out, client.weights().data(), C * sizeof(double));
        memcpy(mean_out, client.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov_out, client.covs().data(), (size_t)C * D * sizeof(double));
    })
}

static MAPCfg make_map_cfg(const char *method, int nbTrainIt, double baggedP, int flags, const double *reg, double alphaMean, const long *norm)
{
    MAPCfg cfg;
    cfg.method = method ? method : "MAPOccDep";
    cfg.nbTrainIt = (unsigned long)nbTrainIt; cfg.baggedFrameProbability = baggedP;
    cfg.meanAdapt = (flags & 1) != 0; cfg.varAdapt = (flags & 2) != 0; cfg.weightAdapt = (flags & 4) != 0;
    if (reg) { cfg.meanReg = reg[0]; cfg.varReg = reg[1]; cfg.weightReg = reg[2]; }
    cfg.meanAlpha = alphaMean;
    if (norm) { cfg.normalizeModel = norm[0] != 0; cfg.normalizeModelMeanOnly = norm[1] != 0; cfg.normalizeModelNbIt = (unsigned long)norm[2]; }
    return cfg;
}

// TrainTarget with every MAPCfg parameter (TrainTools.cpp:95-147): method = MAPAlgo, flags bit 0 / 1 / 2 = meanAdapt / varAdapt /
// weightAdapt, reg[3] = MAPRegFactorMean / Var / Weight, alphaMean = MAPAlphaMean, norm[3] = normalizeModel, MeanOnly, NbIt (nullable)
int liagpu_train_target_ex(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg, int C,
                           const double *w, const double *mean, const double *cov, const char *method, int nbTrainIt, double baggedP,
                           int flags, const double *reg, double alphaMean, const long *norm, double *w_out, double *mean_out, double *cov_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD world = make_mixture(C, D, w, mean, cov);
        MixtureGD client = world;
        adaptModel(fs, segs, world, client, make_map_cfg(method, nbTrainIt, baggedP, flags, reg, alphaMean, norm));
        memcpy(w_out, client.weights().data(), C * sizeof(double));
        memcpy(mean_out, client.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov_out, client.covs().data(), (size_t)C * D * sizeof(double));
    })
}

// computeMAP on its own (TrainTools.cpp:543-556; host arithmetic only -- no device is touched): w / mean / cov = the ML estimate in, the
// adapted model out
int liagpu_compute_map(int C, int D, const double *w0, const double *mean0, const double *cov0, double *w, double *mean, double *cov,
                       double frameCount, const char *method, int flags, const double *reg, double alphaMean)
{
    GUARD({
        MixtureGD init = make_mixture(C, D, w0, mean0, cov0), client = make_mixture(C, D, w, mean, cov);
        computeMAP(init, client, (unsigned long)frameCount, make_map_cfg(method, 1, 1.0, flags, reg, alphaMean, nullptr));
        memcpy(w, client.weights().data(), C * sizeof(double));
        memcpy(mean, client.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, client.covs().data(), (size_t)C * D * sizeof(double));
    })
}

// TopGauss (TopGauss.cpp): compute on the selected frames, write the nbGaussian file, read it back into a fresh object, get() on
// it with `ubm` and (when given) a second model.  out[0] = compute's mean llk, out[1] = get(ubm), out[2] = get(model2),
// out[3] = frames capped; counts_out[T] / snsw_out[T] / snsl_out[T] (nullable) = what was stored; *nbgcnt_out = total entries.
int liagpu_topgauss(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg, int C,
                    const double *w, const double *mean, const double *cov, const double *mean2, double topGauss, int topDistribsCount,
                    int complete, double minLLK, double maxLLK, const char *path, double *out, long *counts_out, long *idx_out,
                    double *snsw_out, double *snsl_out, long *nbgcnt_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        DeviceMixture dubm(srv, ubm);
        TopGauss tg;
        out[0] = tg.compute(dubm, fs, segs, topGauss, topDistribsCount, complete != 0, minLLK, maxLLK);
        out[3] = (double)tg.nbCapped();
        tg.write(path);
        TopGauss back;
        back.read(path);
        if (back.nt() != tg.nt() || back.nbgcnt() != tg.nbgcnt() || back.nbg() != tg.nbg() || back.idx() != tg.idx() || back.snsw() != tg.snsw() ||
            back.snsl() != tg.snsl())
            throw Exception("TopGauss: the file does not read back as written");
        out[1] = back.get(dubm, fs, segs, complete != 0, minLLK, maxLLK);
        out[2] = 0.0;
        if (mean2) {
            MixtureGD m2 = make_mixture(C, D, w, mean2, cov);
            DeviceMixture d2(srv, m2);
            out[2] = back.get(d2, fs, segs, complete != 0, minLLK, maxLLK);
        }
        if (nbgcnt_out) *nbgcnt_out = (long)back.nbgcnt();
        for (unsigned long t = 0; t < back.nt(); ++t) {
            if (counts_out) counts_out[t] = (long)back.nbg()[t];
            if (snsw_out) snsw_out[t] = back.snsw()[t];
            if (snsl_out) snsl_out[t] = back.snsl()[t];
        }
        if (idx_out) for (unsigned long i = 0; i < back.nbgcnt(); ++i) idx_out[i] = (long)back.idx()[i];
    })
}

// ComputeTest for one test file: world + nClients models (same C, D), covariances given as covInv
int liagpu_compute_test(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                        int C, const double *w_world, const double *mean_world, const double *cov_world, int nClients,
                        const double *w_cl, const double *mean_cl, const double *cov_cl, int topDistribsCount,
                        int complete, double minLLK, double maxLLK, int segmentalMode, double *llr_out)
{
    return liagpu_compute_test_timed(device, x, T, D, seg_begin, seg_len, nseg, C, w_world, mean_world, cov_world, nClients, w_cl, mean_cl, cov_cl,
                                     topDistribsCount, complete, minLLK, maxLLK, segmentalMode, llr_out, 1, nullptr);
}

// the same, computeTestLLR run `reps` times on the resident features and models: ms_out[reps] = wall time of each run (bench.py host_layer)
int liagpu_compute_test_timed(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg,
                              int C, const double *w_world, const double *mean_world, const double *cov_world, int nClients,
                              const double *w_cl, const double *mean_cl, const double *cov_cl, int topDistribsCount,
                              int complete, double minLLK, double maxLLK, int segmentalMode, double *llr_out, int reps, double *ms_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD world = make_mixture(C, D, w_world, mean_world, cov_world);
        DeviceMixture dworld(srv, world);
        std::vector<DeviceMixture *> cl;
        const size_t CD = (size_t)C * D;
        for (int i = 0; i < nClients; ++i)
            cl.push_back(new DeviceMixture(srv, make_mixture(C, D, w_cl + (size_t)i * C, mean_cl + i * CD, cov_cl + i * CD)));
        std::vector<double> out;
        try {
            for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
                srv.sync();
                const auto t0 = std::chrono::steady_clock::now();
                out = computeTestLLR(fs, segs, dworld, cl, topDistribsCount, complete != 0, minLLK, maxLLK, segmentalMode != 0);
                srv.sync();
                if (ms_out) ms_out[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
        }
        catch (...) { for (auto p : cl) delete p; throw; }
        for (auto p : cl) delete p;
        memcpy(llr_out, out.data(), out.size() * sizeof(double));
    })
}

// ComputeTest with worldDecime and the WindowLLR mode (ComputeTest.cpp:154-207, UnsupervisedTools.cpp:70-145).
// win_out receives up to max_win rows [idxBegin, idxEnd, llr(client 0..nClients-1)]; *n_win = rows produced.
int liagpu_compute_test_ex(int device, const float *x, long T, int D, const long *seg_begin, const long *seg_len, long nseg, int C,
                           const double *w_world, const double *mean_world, const double *cov_world, int nClients, const double *w_cl,
                           const double *mean_cl, const double *cov_cl, int topDistribsCount, int complete, double minLLK, double maxLLK,
                           int segmentalMode, long worldDecime, long windowSize, long windowDec, double *llr_out, double *win_out,
                           long max_win, long *n_win)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD world = make_mixture(C, D, w_world, mean_world, cov_world);
        DeviceMixture dworld(srv, world);
        std::vector<DeviceMixture *> cl;
        const size_t CD = (size_t)C * D;
        for (int i = 0; i < nClients; ++i)
            cl.push_back(new DeviceMixture(srv, make_mixture(C, D, w_cl + (size_t)i * C, mean_cl + i * CD, cov_cl + i * CD)));
        std::vector<double> out;
        std::vector<WindowOut> wins;
        try {
            out = computeTestLLR(fs, segs, dworld, cl, topDistribsCount, complete != 0, minLLK, maxLLK, segmentalMode != 0,
                                 (unsigned long)worldDecime, (unsigned long)windowSize, (unsigned long)windowDec, &wins);
        } catch (...) { for (auto p : cl) delete p; throw; }
        for (auto p : cl) delete p;
        memcpy(llr_out, out.data(), out.size() * sizeof(double));
        long nw = 0;
        for (const WindowOut &o : wins) {
            if (nw >= max_win) break;
            double *row = win_out + (size_t)nw * (2 + nClients);
            row[0] = (double)o.idxBegin; row[1] = (double)o.idxEnd;
            for (int i = 0; i < nClients; ++i) row[2 + i] = o.llr[i];
            ++nw;
        }
        if (n_win) *n_win = (long)wins.size();
    })
}

// IvExtractor (IvExtractor.cpp:70-148): stats -> substractM -> estimateTETt -> estimateW
int liagpu_iv_extract(int device, const float *x, long T, int D, const long *utt_begin, long U, int C, const double *w,
                      const double *mean, const double *cov, int R, const double *Tmat, double *W_out, double *N_out,
                      double *F_out)
{
    return liagpu_iv_extract_timed(device, x, T, D, utt_begin, U, C, w, mean, cov, R, Tmat, W_out, N_out, F_out, 1, nullptr);
}

// the same, the extraction run `reps` times on the resident features: ms_out[reps x 4] = wall time of computeAndAccumulateTVStat,
// substractM, estimateTETt (first run only: T does not change during extraction, IvExtractor.cpp:136) and estimateW (bench.py host_layer)
int liagpu_iv_extract_timed(int device, const float *x, long T, int D, const long *utt_begin, long U, int C, const double *w,
                            const double *mean, const double *cov, int R, const double *Tmat, double *W_out, double *N_out,
                            double *F_out, int reps, double *ms_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        TVAcc tv(srv, ubm, (unsigned long)R, (unsigned long)U);
        std::vector<SegCluster> lines(U);
        for (long u = 0; u < U; ++u) {
            Seg s; s.begin = (unsigned long)utt_begin[u]; s.length = (unsigned long)(utt_begin[u + 1] - utt_begin[u]); s.source = 0;
            if (s.length) lines[u].push_back(s);
        }
        tv.loadT(std::vector<double>(Tmat, Tmat + (size_t)R * C * D));
        auto lap = [&](std::chrono::steady_clock::time_point &t0) {
            srv.sync();
            const auto t1 = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            t0 = t1;
            return ms;
        };
        for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
            double ms[4] = {0.0, 0.0, 0.0, 0.0};
            srv.sync();
            auto t0 = std::chrono::steady_clock::now();
            tv.computeAndAccumulateTVStat(fs, lines);
            ms[0] = lap(t0);
            if (r == 0) {
                if (N_out) memcpy(N_out, tv.getN().data(), tv.getN().size() * sizeof(double));
                if (F_out) memcpy(F_out, tv.getF().data(), tv.getF().size() * sizeof(double));
                t0 = std::chrono::steady_clock::now();
            }
            tv.substractM();
            ms[1] = lap(t0);
            if (r == 0) { tv.estimateTETt(); ms[2] = lap(t0); }
            tv.estimateW();
            ms[3] = lap(t0);
            if (ms_out) memcpy(ms_out + 4 * r, ms, sizeof(ms));
        }
        memcpy(W_out, tv.getW().data(), tv.getW().size() * sizeof(double));
    })
}

// IvExtractorUbmWeigth / IvExtractorEigenDecomposition (IvExtractor.cpp:150-250, 254-360): the approximate
// extractors.  mode 1 = ubmWeight, 2 = eigenDecomposition.  Q_out [R x R] / D_out [C x R] (mode 2) and Wcov_out
// [R x R] receive the intermediate matrices when given.
int liagpu_iv_extract_approx(int device, int mode, const float *x, long T, int D, const long *utt_begin, long U, int C, const double *w,
                             const double *mean, const double *cov, int R, const double *Tmat, double *W_out, double *Wcov_out,
                             double *Q_out, double *D_out)
{
    GUARD({
        if (mode != 1 && mode != 2) throw Exception("iv_extract_approx: mode must be 1 (ubmWeight) or 2 (eigenDecomposition)");
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        TVAcc tv(srv, ubm, (unsigned long)R, (unsigned long)U);
        std::vector<SegCluster> lines(U);
        for (long u = 0; u < U; ++u) {
            Seg s; s.begin = (unsigned long)utt_begin[u]; s.length = (unsigned long)(utt_begin[u + 1] - utt_begin[u]); s.source = 0;
            if (s.length) lines[u].push_back(s);
        }
        tv.loadT(std::vector<double>(Tmat, Tmat + (size_t)R * C * D));
        tv.normTMatrix();
        std::vector<double> Wc, Q, ev, Dm;
        tv.getWeightedCov(Wc, std::vector<double>(w, w + C));
        if (Wcov_out) memcpy(Wcov_out, Wc.data(), Wc.size() * sizeof(double));
        if (mode == 2) {
            computeEigenProblem(Wc, (unsigned long)R, Q, ev, (unsigned long)R);
            tv.approximateTcTc(Dm, Q);
            if (Q_out) memcpy(Q_out, Q.data(), Q.size() * sizeof(double));
            if (D_out) memcpy(D_out, Dm.data(), Dm.size() * sizeof(double));
        }
        tv.computeAndAccumulateTVStat(fs, lines);
        tv.normStatistics();
        if (mode == 1) tv.estimateWUbmWeight(Wc);
        else { std::fill(tv.getW().begin(), tv.getW().end(), 0.0); tv.estimateWEigenDecomposition(Dm, Q); }
        memcpy(W_out, tv.getW().data(), tv.getW().size() * sizeof(double));
    })
}

// PldaDev::sphericalNuisanceNormalization (PldaTools.cpp:1822-1929): EFR (sph_norm 0) / sphNorm (1) training on a
// development set X [dim x n] (normalised in place); mats [nb_it x dim x dim], means [nb_it x dim].  Optional
// outputs after the last iteration: the WCCN Cholesky factor, the Mahalanobis matrix and lda_rank LDA rows.
int liagpu_backend_train(int device, int dim, long n, double *X, long nspk, const long *sps, int sph_norm, int nb_it, double *mats,
                         double *means, double *wccn, double *mahalanobis, int lda_rank, double *lda)
{
    GUARD({
        GpuServer srv(device);
        PldaDev dev(srv, (unsigned long)dim, std::vector<double>(X, X + (size_t)dim * n), std::vector<unsigned long>(sps, sps + nspk));
        std::vector<std::vector<double> > M, mu;
        dev.sphericalNuisanceNormalization((unsigned long)nb_it, sph_norm != 0, M, mu);
        for (int it = 0; it < nb_it; ++it) {
            memcpy(mats + (size_t)it * dim * dim, M[it].data(), (size_t)dim * dim * sizeof(double));
            memcpy(means + (size_t)it * dim, mu[it].data(), (size_t)dim * sizeof(double));
        }
        memcpy(X, dev.getData().data(), (size_t)dim * n * sizeof(double));
        std::vector<double> t;
        if (wccn) { dev.computeWccnChol(t); memcpy(wccn, t.data(), t.size() * sizeof(double)); }
        if (mahalanobis) { dev.computeMahalanobis(t); memcpy(mahalanobis, t.data(), t.size() * sizeof(double)); }
        if (lda && lda_rank > 0) { dev.computeLDA(t, (unsigned long)lda_rank); memcpy(lda, t.data(), t.size() * sizeof(double)); }
    })
}

// PLDA training (LIA_SpkDet/PLDA/src/PLDA.cpp:80-95): nb_it EM iterations from given F, G, Sigma; X is left centred by
// the accumulated minimum-divergence shifts like the reference's _Dev.
int liagpu_plda_train(int device, int dim, long n, double *X, long nspk, const long *sps, int rf, int rg, int nb_it, double *F, double *G,
                      double *Sigma, double *Delta, double *original_mean)
{
    GUARD({
        GpuServer srv(device);
        PldaDev dev(srv, (unsigned long)dim, std::vector<double>(X, X + (size_t)dim * n), std::vector<unsigned long>(sps, sps + nspk));
        PldaModel plda(dev, (unsigned long)rf, (unsigned long)rg, std::vector<double>(F, F + (size_t)dim * rf),
                       std::vector<double>(G, G + (size_t)dim * rg), std::vector<double>(Sigma, Sigma + (size_t)dim * dim));
        for (int it = 0; it < nb_it; ++it) plda.em_iteration();
        memcpy(F, plda.getF().data(), plda.getF().size() * sizeof(double));
        memcpy(G, plda.getG().data(), plda.getG().size() * sizeof(double));
        memcpy(Sigma, plda.getSigma().data(), plda.getSigma().size() * sizeof(double));
        memcpy(Delta, plda.getDelta().data(), plda.getDelta().size() * sizeof(double));
        if (original_mean) memcpy(original_mean, plda.getOriginalMean().data(), (size_t)dim * sizeof(double));
        memcpy(X, dev.getData().data(), (size_t)dim * n * sizeof(double));
    })
}

// IvTest (IvTest.cpp:73-471).  scoring: 0 cosine, 1 mahalanobis, 2 2cov, 3 plda.  dev [dim x nDev] with sessions grouped by
// speaker; enrol [dim x nEnrol] grouped by model; test [dim x nTest]; scores [nModels x nTest].
int liagpu_iv_test(int device, int dim, long nDev, const double *dev, long nspk, const long *sps, long nModels, const long *enrolPerModel,
                   const double *enrol, long nTest, const double *test, int ivNorm, int ivNormIt, int sphNorm, int lda, int ldaRank,
                   int wccn, int scoring, int rf, int rg, int pldaIt, const double *F, const double *G, const double *Sigma, double *scores)
{
    GUARD({
        static const char *names[] = {"cosine", "mahalanobis", "2cov", "plda"};
        if (scoring < 0 || scoring > 3) throw Exception("Scoring option is invalid, must be: cosine OR mahalanobis OR 2cov OR plda");
        GpuServer srv(device);
        PldaDev pd(srv, (unsigned long)dim, std::vector<double>(dev, dev + (size_t)dim * nDev), std::vector<unsigned long>(sps, sps + nspk));
        IvTestCfg cfg;
        cfg.ivNorm = ivNorm != 0; cfg.ivNormIterationNb = (unsigned long)ivNormIt; cfg.sphNorm = sphNorm != 0;
        cfg.LDA = lda != 0; cfg.ldaRank = (unsigned long)ldaRank; cfg.WCCN = wccn != 0; cfg.scoring = names[scoring];
        cfg.pldaRankF = (unsigned long)rf; cfg.pldaRankG = (unsigned long)rg; cfg.pldaNbIt = (unsigned long)pldaIt;
        std::vector<unsigned long> epm(enrolPerModel, enrolPerModel + nModels);
        unsigned long nEnrol = 0;
        for (unsigned long e : epm) nEnrol += e;
        const unsigned long d2 = cfg.ivNorm && cfg.LDA ? 0 : 0; (void)d2;
        std::vector<double> f, g, s;
        if (scoring == 3) {
            // the PLDA matrices live in the space AFTER normalisation / LDA: their leading dimension is that space's
            const unsigned long pd_dim = cfg.ivNorm && cfg.LDA ? cfg.ldaRank : (unsigned long)dim;
            f.assign(F, F + pd_dim * rf); g.assign(G, G + pd_dim * rg); s.assign(Sigma, Sigma + pd_dim * pd_dim);
        }
        std::vector<double> out = ivTest(srv, cfg, pd, std::vector<double>(enrol, enrol + (size_t)dim * nEnrol), epm,
                                         std::vector<double>(test, test + (size_t)dim * nTest), (unsigned long)nTest, f, g, s);
        memcpy(scores, out.data(), out.size() * sizeof(double));
    })
}

// TotalVariability (TotalVariability.cpp:118-169): nbIt iterations on precomputed N, F
int liagpu_tv_train(int device, long U, int C, int D, const double *w, const double *mean, const double *cov, int R,
                    const double *N, const double *F, double *Tmat, int nbIt, int minDiv, double *mean_out)
{
    GUARD({
        GpuServer srv(device);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        TVAcc tv(srv, ubm, (unsigned long)R, (unsigned long)U);
        tv.loadT(std::vector<double>(Tmat, Tmat + (size_t)R * C * D));
        tv.setStats(N, F);         // one upload; everything below stays on the device until getT()
        tv.storeStats();
        for (int it = 0; it < nbIt; ++it) {
            if (it) tv.restoreStatsAndSubstractM(); // the reference reloads N and F from disk every iteration (:152-153) and centres
            else tv.substractM();                   // them in place: one device pass over F here
            tv.estimateTETt();
            tv.estimateAandC();
            tv.updateTestimate();
            if (minDiv) tv.minDivergence();
        }
        memcpy(Tmat, tv.getT().data(), tv.getT().size() * sizeof(double));
        if (mean_out) memcpy(mean_out, tv.getUbmMeans().data(), tv.getUbmMeans().size() * sizeof(double));
    })
}

// TVAcc::initT ("normal" law) for a UBM with the given inverse variances: seeds glibc with `seed` first (the reference never calls
// srand, i.e. seed 1) and fills Tmat [R x C*D]
int liagpu_tv_init_t(int device, int C, int D, const double *w, const double *mean, const double *cov, int R, unsigned seed, double *Tmat)
{
    GUARD({
        GpuServer srv(device);
        TVAcc tv(srv, make_mixture(C, D, w, mean, cov), (unsigned long)R, 1);
        srand(seed);
        tv.initT("normal");
        memcpy(Tmat, tv.getT().data(), (size_t)R * C * D * sizeof(double));
    })
}

// computeAndAccumulateTVStat with the file -> ndx-line map: file f = frames [file_begin[f], file_begin[f+1]) of x, line l lists the
// files line_files[line_off[l] .. line_off[l+1]); N [nlines x C], F [nlines x C*D]
int liagpu_tv_stats_lines(int device, const float *x, long T, int D, const long *file_begin, long nfiles, long nlines, const long *line_off,
                          const long *line_files, int C, const double *w, const double *mean, const double *cov, double *N, double *F)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        TVAcc tv(srv, make_mixture(C, D, w, mean, cov), 1, (unsigned long)nlines);
        std::vector<SegCluster> segs(nfiles);
        for (long f = 0; f < nfiles; ++f) { Seg s; s.begin = (unsigned long)file_begin[f]; s.length = (unsigned long)(file_begin[f + 1] - file_begin[f]); segs[f].push_back(s); }
        std::vector<std::vector<unsigned long> > lines(nlines);
        for (long l = 0; l < nlines; ++l)
            for (long p = line_off[l]; p < line_off[l + 1]; ++p) lines[l].push_back((unsigned long)line_files[p]);
        tv.computeAndAccumulateTVStat(fs, segs, lines);
        memcpy(N, tv.getN().data(), tv.getN().size() * sizeof(double));
        memcpy(F, tv.getF().data(), tv.getF().size() * sizeof(double));
    })
}

// One rank of a multi-GPU TotalVariability run (configs[3]: utterances sharded over the GPUs of a node, one process -- or host
// thread -- per GPU).  N, F: the statistics of THIS rank's U utterances; Tmat: the same initial matrix on every rank.  The
// ranks meet through `id_file` (rank 0 publishes the RCCL id there, gmmiv_comm_exchange_id_file); per iteration the only
// exchange is reduce-scatter(A, Cmx) / all-gather(T) / all-reduce(R, r, meanW) on device buffers (TVAcc::updateTestimate(comm)).
// U_total = utterances over all ranks (the session count of minDivergence).  times_ms (nullable, 4 x nbIt): wall-clock of
// estimateTETt / estimateAandC / updateTestimate (with its collectives) / minDivergence per iteration.
int liagpu_tv_train_dist2(int device, int world, int rank, const char *id_file, long U, long U_total, int C, int D, const double *w,
                          const double *mean, const double *cov, int R, const double *N, const double *F, double *Tmat, int nbIt,
                          int minDiv, int overlap, double *mean_out, double *times_ms);
int liagpu_tv_train_dist(int device, int world, int rank, const char *id_file, long U, long U_total, int C, int D, const double *w,
                         const double *mean, const double *cov, int R, const double *N, const double *F, double *Tmat, int nbIt,
                         int minDiv, double *mean_out, double *times_ms)
{
    return liagpu_tv_train_dist2(device, world, rank, id_file, U, U_total, C, D, w, mean, cov, R, N, F, Tmat, nbIt, minDiv, 0, mean_out, times_ms);
}
// the same with the exchange started early when overlap != 0 (TVAcc::setOverlap: the reduce-scatter of A from inside estimateAandC,
// the all-gather of T joined inside minDivergence); bitwise the results of overlap == 0
int liagpu_tv_train_dist2(int device, int world, int rank, const char *id_file, long U, long U_total, int C, int D, const double *w,
                          const double *mean, const double *cov, int R, const double *N, const double *F, double *Tmat, int nbIt,
                          int minDiv, int overlap, double *mean_out, double *times_ms)
{
    GUARD({
        GpuServer srv(device);
        gmmiv_comm *comm = nullptr;
        if (world > 1) {
            unsigned char id[GMMIV_COMM_ID_BYTES];
            srv.check(gmmiv_comm_exchange_id_file(id_file, rank, id, 300.0));
            srv.check(gmmiv_comm_create(srv.ctx(), world, rank, id, &comm));
        }
        struct CommGuard { gmmiv_comm *c; ~CommGuard() { gmmiv_comm_destroy(c); } } guard{comm};
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        TVAcc tv(srv, ubm, (unsigned long)R, (unsigned long)U);
        tv.loadT(std::vector<double>(Tmat, Tmat + (size_t)R * C * D));
        tv.setStats(N, F);
        tv.storeStats();
        if (overlap) tv.setOverlap(comm);
        auto now = [&]() { srv.check(gmmiv_ctx_sync(srv.ctx())); return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        for (int it = 0; it < nbIt; ++it) {
            if (it) tv.restoreStatsAndSubstractM();
            else tv.substractM();
            auto t0 = now();
            tv.estimateTETt();
            auto t1 = now();
            tv.estimateAandC();
            auto t2 = now();
            tv.updateTestimate(comm, (unsigned long)U_total);
            auto t3 = now();
            if (minDiv) tv.minDivergence();
            else tv.finishT();
            auto t4 = now();
            if (times_ms) { times_ms[4 * it] = ms(t0, t1); times_ms[4 * it + 1] = ms(t1, t2); times_ms[4 * it + 2] = ms(t2, t3); times_ms[4 * it + 3] = ms(t3, t4); }
        }
        memcpy(Tmat, tv.getT().data(), tv.getT().size() * sizeof(double));
        if (mean_out) memcpy(mean_out, tv.getUbmMeans().data(), tv.getUbmMeans().size() * sizeof(double));
    })
}

// One rank of a multi-GPU TrainWorld run: x = this rank's frames, global_cov = the variance-control reference (replicated);
// per iteration ONE gmmiv_allreduce_f64 of the flat EM accumulator on the device (trainModelStream(..., comm)).
int liagpu_train_world_dist(int device, int world, int rank, const char *id_file, const float *x, long T, int D, const long *seg_begin,
                            const long *seg_len, long nseg, int C, double *w, double *mean, double *cov, int nbTrainIt,
                            double initVarFloor, double finalVarFloor, double initVarCeil, double finalVarCeil, const double *global_cov,
                            double *llk_it_out)
{
    GUARD({
        GpuServer srv(device);
        gmmiv_comm *comm = nullptr;
        unsigned char id[GMMIV_COMM_ID_BYTES] = {0};
        if (world > 1) srv.check(gmmiv_comm_exchange_id_file(id_file, rank, id, 300.0));
        srv.check(gmmiv_comm_create(srv.ctx(), world, rank, id, &comm));
        struct CommGuard { gmmiv_comm *c; ~CommGuard() { gmmiv_comm_destroy(c); } } guard{comm};
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        MixtureGD model = make_mixture(C, D, w, mean, cov);
        TrainCfg cfg;
        cfg.nbTrainIt = nbTrainIt;
        cfg.initVarianceFlooring = initVarFloor; cfg.finalVarianceFlooring = finalVarFloor;
        cfg.initVarianceCeiling = initVarCeil; cfg.finalVarianceCeiling = finalVarCeil;
        std::vector<double> llk = trainModelStream(cfg, fs, segs, std::vector<double>(global_cov, global_cov + D), model, comm);
        memcpy(w, model.weights().data(), C * sizeof(double));
        memcpy(mean, model.means().data(), (size_t)C * D * sizeof(double));
        memcpy(cov, model.covs().data(), (size_t)C * D * sizeof(double));
        if (llk_it_out) memcpy(llk_it_out, llk.data(), llk.size() * sizeof(double));
    })
}

// EigenVoice / EigenChannel / EstimateDMatrix on given statistics (task 0 / 1 / 2; EigenVoice.cpp:71-160,
// EigenChannel.cpp:71-165, EstimateDMatrix.cpp:103-210 minus the file I/O).  V, U, Dm: initial matrices in, trained out
// (only the task's own matrix changes); Y, X, Z: the factors of the last estimate (outputs, may be NULL).
int liagpu_jfa_train(int device, int task, long nspk, const long *sessPerSpk, int C, int D, const double *w, const double *mean,
                     const double *cov, int rankEV, int rankEC, const double *N, const double *N_h, const double *F_X, const double *F_X_h,
                     double *V, double *U, double *Dm, const double *Z0, int nbIt, int orthoV, double *Y, double *X, double *Z)
{
    GUARD({
        GpuServer srv(device);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        std::vector<unsigned long> sps(sessPerSpk, sessPerSpk + nspk);
        JFAAcc jfa(srv, ubm, (unsigned long)rankEV, (unsigned long)rankEC, sps);
        const size_t SV = (size_t)C * D, nsess = jfa.getNSessions();
        jfa.setStats(std::vector<double>(N, N + (size_t)nspk * C), std::vector<double>(N_h, N_h + nsess * C),
                     std::vector<double>(F_X, F_X + (size_t)nspk * SV), std::vector<double>(F_X_h, F_X_h + nsess * SV));
        jfa.loadEV(std::vector<double>(V, V + (size_t)rankEV * SV));
        jfa.loadEC(std::vector<double>(U, U + (size_t)rankEC * SV));
        jfa.loadD(std::vector<double>(Dm, Dm + SV));
        if (Z0) jfa.getZ().assign(Z0, Z0 + (size_t)nspk * SV);
        if (task == 0) eigenVoice(jfa, (unsigned long)nbIt, orthoV != 0);
        else if (task == 1) eigenChannel(jfa, (unsigned long)nbIt);
        else if (task == 2) estimateDMatrix(jfa, (unsigned long)nbIt);
        else throw Exception("liagpu_jfa_train: task must be 0 (EigenVoice), 1 (EigenChannel) or 2 (EstimateDMatrix)");
        memcpy(V, jfa.getV().data(), jfa.getV().size() * sizeof(double));
        memcpy(U, jfa.getU().data(), jfa.getU().size() * sizeof(double));
        memcpy(Dm, jfa.getD().data(), SV * sizeof(double));
        if (Y) memcpy(Y, jfa.getY().data(), jfa.getY().size() * sizeof(double));
        if (X) memcpy(X, jfa.getX().data(), jfa.getX().size() * sizeof(double));
        if (Z) memcpy(Z, jfa.getZ().data(), jfa.getZ().size() * sizeof(double));
    })
}

// ComputeTest, JFA dot-product scoring (ComputeTest.cpp:303-358) for nTest segments given their statistics (N == N_h, F == F_h:
// one session each) and the client supervectors [nClients x SV]; scores [nTest x nClients]
int liagpu_jfa_dot_product(int device, long nTest, int C, int D, const double *w, const double *mean, const double *cov, int rankEV,
                           int rankEC, const double *N, const double *F, const double *V, const double *U, const double *Dm,
                           long nClients, const double *clientSV, double *scores)
{
    GUARD({
        GpuServer srv(device);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        std::vector<unsigned long> sps((size_t)nTest, 1ul);
        JFAAcc jfa(srv, ubm, (unsigned long)rankEV, (unsigned long)rankEC, sps);
        const size_t SV = (size_t)C * D;
        const std::vector<double> Nv(N, N + (size_t)nTest * C), Fv(F, F + (size_t)nTest * SV);
        jfa.setStats(Nv, Nv, Fv, Fv);
        jfa.loadEV(std::vector<double>(V, V + (size_t)rankEV * SV));
        jfa.loadEC(std::vector<double>(U, U + (size_t)rankEC * SV));
        jfa.loadD(std::vector<double>(Dm, Dm + SV));
        std::vector<double> sc = computeTestDotProduct(srv, jfa, std::vector<double>(clientSV, clientSV + (size_t)nClients * SV), (unsigned long)nClients);
        memcpy(scores, sc.data(), sc.size() * sizeof(double));
    })
}

// Baum-Welch statistics of a TVAcc that lives on ONE server from the frames of a FeatureBuffer that lives on ANOTHER (the
// reference hands any FeatureServer to any accumulator, AccumulateTVStat.cpp:281-351).  The accumulator's server also owns a clean
// buffer (x_own): the screening decision of the call must follow the buffer that is READ -- a dirty x there must not poison N / F.
// unusable_out[2] = unusable frames of (the accumulator's own buffer, the foreign buffer) as counted at upload.
int liagpu_tv_stats_cross_server(int device, const float *x_own, long T_own, const float *x, long T, int D, const long *utt_begin, long U,
                                 int C, const double *w, const double *mean, const double *cov, double *N_out, double *F_out,
                                 long *unusable_out, long *assume_finite_after)
{
    GUARD({
        GpuServer srvAcc(device), srvFeat(device);
        FeatureBuffer own(srvAcc, x_own, (unsigned long)T_own, (unsigned long)D);
        FeatureBuffer fs(srvFeat, x, (unsigned long)T, (unsigned long)D);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        TVAcc tv(srvAcc, ubm, 2, (unsigned long)U);
        std::vector<SegCluster> lines(U);
        for (long u = 0; u < U; ++u) {
            Seg s; s.begin = (unsigned long)utt_begin[u]; s.length = (unsigned long)(utt_begin[u + 1] - utt_begin[u]); s.source = 0;
            if (s.length) lines[u].push_back(s);
        }
        tv.computeAndAccumulateTVStat(fs, lines);
        memcpy(N_out, tv.getN().data(), tv.getN().size() * sizeof(double));
        memcpy(F_out, tv.getF().data(), tv.getF().size() * sizeof(double));
        if (unusable_out) { unusable_out[0] = (long)own.unusableFrames(); unusable_out[1] = (long)fs.unusableFrames(); }
        // the option is the user's again once the call has returned (it was 0 before: never set)
        if (assume_finite_after) *assume_finite_after = gmmiv_ctx_set_option(srvAcc.ctx(), "assume_finite", 0);
    })
}

// JFA statistics from frames (JFAAcc::computeAndAccumulateJFAStat, :515-577): sessions = utterance ranges, grouped by speaker
int liagpu_jfa_stats(int device, const float *x, long T, int D, const long *sess_begin, long nspk, const long *sessPerSpk, int C,
                     const double *w, const double *mean, const double *cov, double *N, double *N_h, double *F_X, double *F_X_h)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, x, (unsigned long)T, (unsigned long)D);
        MixtureGD ubm = make_mixture(C, D, w, mean, cov);
        std::vector<unsigned long> sps(sessPerSpk, sessPerSpk + nspk);
        JFAAcc jfa(srv, ubm, 1, 1, sps);
        std::vector<SegCluster> segs(jfa.getNSessions());
        for (unsigned long h = 0; h < jfa.getNSessions(); ++h) {
            Seg s; s.begin = (unsigned long)sess_begin[h]; s.length = (unsigned long)(sess_begin[h + 1] - sess_begin[h]); s.source = 0;
            segs[h].push_back(s);
        }
        jfa.computeAndAccumulateJFAStat(fs, segs);
        memcpy(N, jfa.getN().data(), jfa.getN().size() * sizeof(double));
        memcpy(N_h, jfa.getN_h().data(), jfa.getN_h().size() * sizeof(double));
        memcpy(F_X, jfa.getF_X().data(), jfa.getF_X().size() * sizeof(double));
        memcpy(F_X_h, jfa.getF_X_h().data(), jfa.getF_X_h().size() * sizeof(double));
    })
}

// ComputeTest from FILES for one ndx line (test file + client list): RAW models, .prm features with
// featureServerMask, .lbl selection.  Writes the NIST-style result lines (segmental mode) into out_text
// and the LLRs into llr_out[nseg x nClients].  ComputeTest.cpp:129-215 + the format readers of io.h.
int liagpu_compute_test_files(int device, const char *world_path, int nClients, const char **client_paths,
                              const char **client_names, const char *prm_path, const char *lbl_path, const char *mask,
                              const char *label, double frameLength, int topDistribsCount, int complete, double minLLK,
                              double maxLLK, const char *gender, const char *testName, double threshold, double *llr_out,
                              long max_llr, char *out_text, long out_cap)
{
    GUARD({
        GpuServer srv(device);
        MixtureGD world = readMixtureRAW(world_path);
        FeatureFile ff = readFeatureFile(prm_path, mask ? mask : "");
        if (ff.vectSize != world.getVectSize()) throw Exception("vectSize of features and world model differ");
        FeatureBuffer fs(srv, ff.data.data(), ff.nFrames, ff.vectSize);
        SegCluster segs = selectSegments(readLabelFile(lbl_path), label, frameLength);
        DeviceMixture dworld(srv, world);
        std::vector<DeviceMixture *> cl;
        std::vector<double> out;
        try {
            for (int i = 0; i < nClients; ++i) cl.push_back(new DeviceMixture(srv, readMixtureRAW(client_paths[i])));
            out = computeTestLLR(fs, segs, dworld, cl, topDistribsCount, complete != 0, minLLK, maxLLK, true);
        } catch (...) { for (auto p : cl) delete p; throw; }
        for (auto p : cl) delete p;
        if ((long)out.size() > max_llr) throw Exception("llr_out too small");
        memcpy(llr_out, out.data(), out.size() * sizeof(double));
        std::string text;
        for (size_t s = 0; s < segs.size(); ++s)
            for (int i = 0; i < nClients; ++i)   // frameIdxToTime(begin), frameIdxToTime(begin + length) (ComputeTest.cpp:186)
                text += resultLine(out[s * nClients + i], client_names[i], testName, gender, threshold, true,
                                   frameIdxToTime(segs[s].begin, frameLength), frameIdxToTime(segs[s].begin + segs[s].length, frameLength)) + "\n";
        if ((long)text.size() + 1 > out_cap) throw Exception("out_text too small");
        memcpy(out_text, text.c_str(), text.size() + 1);
    })
}

// EnergyDetector on one energy column (EnergyDetector.cpp:190-285, thresholdMode meanStd): energy [T] float32 (featureServerMask picks
// the column, vectSize 1), selected segments; out_begin / out_len [max_out] <- the output segments (frames of the SELECTION), model_out
// [3 * C] <- weights | means | covariances, *threshold_out <- the threshold.
int liagpu_energy_detector(int device, const float *energy, long T, const long *seg_begin, const long *seg_len, long nseg, int C, int nbTrainIt,
                           double varianceFlooring, double varianceCeiling, double alpha, long *out_begin, long *out_len, long max_out,
                           long *n_out, double *model_out, double *threshold_out)
{
    GUARD({
        GpuServer srv(device);
        FeatureBuffer fs(srv, energy, (unsigned long)T, 1);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg);
        EnergyDetectorCfg cfg;
        cfg.nbTrainIt = (unsigned long)nbTrainIt; cfg.mixtureDistribCount = (unsigned long)C;
        cfg.varianceFlooring = varianceFlooring; cfg.varianceCeiling = varianceCeiling; cfg.alpha = alpha;
        MixtureGD model(C, 1);
        double th = 0.0;
        SegCluster out = energyDetector(fs, segs, cfg, &model, &th);
        if ((long)out.size() > max_out) throw Exception("out_begin too small");
        for (size_t i = 0; i < out.size(); ++i) { out_begin[i] = (long)out[i].begin; out_len[i] = (long)out[i].length; }
        *n_out = (long)out.size();
        if (model_out)
            for (int c = 0; c < C; ++c) { model_out[c] = model.weight(c); model_out[C + c] = model.getMean(c, 0); model_out[2 * C + c] = model.getCov(c, 0); }
        if (threshold_out) *threshold_out = th;
    })
}

// selectFrames alone (EnergyDetector.cpp:118-157; host only, no GPU): energy [T], threshold, selected segments -> output segments
int liagpu_select_frames(const float *energy, long T, double threshold, const long *seg_begin, const long *seg_len, long nseg, long *out_begin,
                         long *out_len, long max_out, long *n_out, long *count_out)
{
    GUARD({
        std::vector<float> e(energy, energy + T);
        SegCluster segs = make_cluster(seg_begin, seg_len, nseg), out;
        const unsigned long cnt = selectFrames(e, threshold, segs, out);
        if ((long)out.size() > max_out) throw Exception("out_begin too small");
        for (size_t i = 0; i < out.size(); ++i) { out_begin[i] = (long)out[i].begin; out_len[i] = (long)out[i].length; }
        *n_out = (long)out.size();
        if (count_out) *count_out = (long)cnt;
    })
}

// GmmTokenizer driven from its files (LIA_Utils/GmmTokenizer/src/GmmTokenizer.cpp:169-207 symbols, :120-164 confusion matrix): RAW world
// model, .prm features (masked), .lbl segments with the selected label.  symbols_out [max_symbols] <- one best-Gaussian index per selected
// frame (*n_symbols of them; skipped when symbols_out is NULL); confusion_out [C x C] (nullable, zeroed here) <- the nBest = topDistribsCount
// confusion counts; matrix_path (nullable / "") <- the same matrix as a DT file, like mce_matrix.save (:160).
int liagpu_gmm_tokenizer_files(int device, const char *world_path, const char *prm_path, const char *lbl_path, const char *mask,
                               const char *label, double frameLength, int topDistribsCount, double minLLK, double maxLLK,
                               long *symbols_out, long max_symbols, long *n_symbols, long *confusion_out, long *dims,
                               const char *matrix_path)
{
    GUARD({
        GpuServer srv(device);
        MixtureGD world = readMixtureRAW(world_path);
        FeatureFile ff = readFeatureFile(prm_path, mask ? mask : "");
        if (ff.vectSize != world.getVectSize()) throw Exception("vectSize of features and world model differ");
        FeatureBuffer fs(srv, ff.data.data(), ff.nFrames, ff.vectSize);
        SegCluster segs = selectSegments(readLabelFile(lbl_path), label, frameLength);
        DeviceMixture dworld(srv, world);
        const unsigned long C = world.getDistribCount();
        if (dims) { dims[0] = (long)C; dims[1] = (long)world.getVectSize(); dims[2] = (long)totalFrame(segs); }
        if (symbols_out) {
            std::vector<unsigned long> stream;
            computeSymbols(segs, fs, dworld, stream, topDistribsCount, minLLK, maxLLK);
            if ((long)stream.size() > max_symbols) throw Exception("symbols_out too small");
            for (size_t i = 0; i < stream.size(); ++i) symbols_out[i] = (long)stream[i];
            if (n_symbols) *n_symbols = (long)stream.size();
        }
        if (confusion_out || (matrix_path && *matrix_path)) {
            std::vector<unsigned long> mce((size_t)C * C, 0);
            computeConfusionMatrix(segs, fs, dworld, (unsigned long)topDistribsCount, mce, minLLK, maxLLK);
            if (confusion_out) for (size_t i = 0; i < mce.size(); ++i) confusion_out[i] = (long)mce[i];
            if (matrix_path && *matrix_path) {
                MatrixD m; m.rows = C; m.cols = C; m.v.assign(mce.begin(), mce.end());
                writeMatrixDT(matrix_path, m);
            }
        }
    })
}

// XML mixture round trip + DB / DT matrices + per-id vector files (CPU tests; no GPU involved).
// xml_in -> xml_out (re-written) and raw_out (the same model as a RAW file); dims = {C, D}; first = {weight 0, covInv(0,0), mean(0,0)}
int liagpu_io_xml(const char *xml_in, const char *xml_out, const char *raw_out, long *dims, double *first)
{
    GUARD({
        MixtureGD m = readMixture(xml_in);
        dims[0] = (long)m.getDistribCount(); dims[1] = (long)m.getVectSize();
        first[0] = m.weight(0); first[1] = m.getCovInv(0, 0); first[2] = m.getMean(0, 0);
        writeMixtureXML(xml_out, m);
        if (raw_out && *raw_out) {
            writeMixtureRAW(raw_out, m);
            MixtureGD r = readMixture(raw_out);        // sniffed as RAW
            if (r.getDistribCount() != m.getDistribCount() || r.getMean(0, 0) != m.getMean(0, 0)) throw Exception("RAW re-read differs");
        }
    })
}
// width of the extents writeMatrixDB puts in a DB header (4 or 8 bytes; anything else only queries); returns the previous one
int liagpu_io_db_header_bytes(int bytes) { return setMatrixDBHeaderBytes(bytes); }
// matrix file -> matrix file in another format; dims = {rows, cols}
int liagpu_io_matrix_convert(const char *in, const char *fmt_in, const char *out, const char *fmt_out, long *dims)
{
    GUARD({
        const MatrixD m = readMatrix(in, fmt_in);
        dims[0] = (long)m.rows; dims[1] = (long)m.cols;
        writeMatrix(out, m, fmt_out);
    })
}
// W [n x rank] -> one vector file per id under dir (saveWbyFile), read back as columns (PldaTest::load): back [rank x n]
int liagpu_io_vectors(const char *dir, int n, const char **ids, const char *ext, const char *fmt, int rank, const double *W, double *back)
{
    GUARD({
        std::vector<std::string> v(ids, ids + n);
        saveVectorsById(std::string(dir) + "/", v, ext, std::vector<double>(W, W + (size_t)n * rank), (unsigned long)rank, fmt);
        unsigned long dim = 0;
        const std::vector<double> cols = loadVectorsById(dir, v, ext, dim, fmt);
        if (dim != (unsigned long)rank) throw Exception("vector files: dimension changed on the way");
        memcpy(back, cols.data(), cols.size() * sizeof(double));
    })
}

// format round trips used by the CPU tests (no GPU involved)
int liagpu_io_roundtrip(const char *raw_in, const char *raw_out, const char *prm_in, const char *prm_out, const char *mask,
                        long *dims /* C, D, nFrames, vectSize */, double *first_mean, float *first_frame)
{
    GUARD({
        MixtureGD m = readMixtureRAW(raw_in);
        writeMixtureRAW(raw_out, m);
        FeatureFile f = readFeatureFile(prm_in, "");
        writeFeatureFile(prm_out, f);
        FeatureFile fm = readFeatureFile(prm_in, mask);
        dims[0] = (long)m.getDistribCount(); dims[1] = (long)m.getVectSize(); dims[2] = (long)fm.nFrames; dims[3] = (long)fm.vectSize;
        for (unsigned long d = 0; d < m.getVectSize(); ++d) first_mean[d] = m.getMean(0, d);
        for (unsigned long d = 0; d < fm.vectSize; ++d) first_frame[d] = fm.data[d];
    })
}

int liagpu_label_segments(const char *lbl_path, const char *label, double frameLength, long *begin, long *len, long cap, long *n)
{
    GUARD({
        SegCluster c = selectSegments(readLabelFile(lbl_path), label, frameLength);
        if ((long)c.size() > cap) throw Exception("segment buffer too small");
        for (size_t i = 0; i < c.size(); ++i) { begin[i] = (long)c[i].begin; len[i] = (long)c[i].length; }
        *n = (long)c.size();
    })
}

} // extern "C"
