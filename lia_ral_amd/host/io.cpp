// io.cpp -- see io.h
#include "io.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <sstream>

namespace liagpu {

static std::vector<unsigned char> slurp(const std::string &path)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) throw Exception("cannot open file [" + path + "]");
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::vector<int> parseFeatureMask(const std::string &mask)
{
    std::vector<int> cols;
    std::stringstream ss(mask);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        if (tok.empty()) continue;
        const size_t dash = tok.find('-');
        int a, b;
        if (dash == std::string::npos) a = b = atoi(tok.c_str());
        else { a = atoi(tok.substr(0, dash).c_str()); b = atoi(tok.substr(dash + 1).c_str()); }
        if (a < 0 || b < a) throw Exception("bad featureServerMask [" + mask + "]");
        for (int c = a; c <= b; ++c) cols.push_back(c);
    }
    return cols;
}

FeatureFile readFeatureFile(const std::string &path, const std::string &mask)
{
    const std::vector<unsigned char> b = slurp(path);
    if (b.size() < 16) throw Exception("feature file too short [" + path + "]");
    uint32_t h[4];
    memcpy(h, b.data(), 16);
    FeatureFile f;
    f.baseDim = h[1]; f.flags = h[3];
    const unsigned long n = h[2];
    if (n == 0) return f;
    if ((b.size() - 16) % (4 * n)) throw Exception("feature file size does not match its frame count [" + path + "]");
    const unsigned long dim = (b.size() - 16) / (4 * n);
    std::vector<int> cols = parseFeatureMask(mask);
    if (cols.empty()) for (unsigned long c = 0; c < dim; ++c) cols.push_back((int)c);
    for (int c : cols) if ((unsigned long)c >= dim) throw Exception("featureServerMask selects a column beyond vectSize");
    f.nFrames = n; f.vectSize = cols.size();
    f.data.resize(n * cols.size());
    const float *src = (const float *)(b.data() + 16);
    for (unsigned long t = 0; t < n; ++t)
        for (size_t k = 0; k < cols.size(); ++k) f.data[t * cols.size() + k] = src[t * dim + cols[k]];
    return f;
}

void writeFeatureFile(const std::string &path, const FeatureFile &f)
{
    std::ofstream o(path.c_str(), std::ios::binary);
    if (!o) throw Exception("cannot write [" + path + "]");
    const uint32_t h[4] = {2u, f.baseDim, (uint32_t)f.nFrames, f.flags};
    o.write((const char *)h, 16);
    o.write((const char *)f.data.data(), f.data.size() * sizeof(float));
}

std::vector<LabelSeg> readLabelFile(const std::string &path)
{
    std::ifstream f(path.c_str());
    if (!f) throw Exception("cannot open label file [" + path + "]");
    std::vector<LabelSeg> out;
    std::string line;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        LabelSeg s;
        if (ss >> s.begin_s >> s.end_s >> s.label) out.push_back(s);
    }
    return out;
}

SegCluster selectSegments(const std::vector<LabelSeg> &lab, const std::string &labelSelectedFrames, double frameLength,
                          unsigned long source)
{
    SegCluster c;
    for (const LabelSeg &l : lab)
        if (l.label == labelSelectedFrames) c.push_back(segFromLabel(l.begin_s, l.end_s, frameLength, source));
    return c;
}

MixtureGD readMixtureRAW(const std::string &path)
{
    const std::vector<unsigned char> b = slurp(path);
    if (b.size() < 8) throw Exception("mixture file too short [" + path + "]");
    uint32_t C, D;
    memcpy(&C, b.data(), 4);
    memcpy(&D, b.data() + 4, 4);
    const size_t rec = 17 + 16 * (size_t)D, want = 8 + 8 * (size_t)C + C * rec;
    if (C == 0 || D == 0 || b.size() != want) {
        char msg[256];
        snprintf(msg, sizeof(msg), "RAW mixture [%s]: %zu bytes, expected %zu for %u x %u (text-mode damaged file?)", path.c_str(),
                 b.size(), want, C, D);
        throw Exception(msg);
    }
    MixtureGD m(C, D);
    memcpy(m.weights().data(), b.data() + 8, 8 * (size_t)C);
    for (uint32_t c = 0; c < C; ++c) {
        const unsigned char *p = b.data() + 8 + 8 * (size_t)C + c * rec + 17; // skip cst, det, flag (recomputed)
        std::vector<double> civ(D), mu(D);
        memcpy(civ.data(), p, 8 * (size_t)D);
        memcpy(mu.data(), p + 8 * (size_t)D, 8 * (size_t)D);
        for (uint32_t d = 0; d < D; ++d) { m.setMean(c, mu[d], d); m.setCov(c, 1.0 / civ[d], d); }
    }
    m.computeAll();
    return m;
}

void writeMixtureRAW(const std::string &path, const MixtureGD &m)
{
    std::ofstream o(path.c_str(), std::ios::binary);
    if (!o) throw Exception("cannot write [" + path + "]");
    const uint32_t C = (uint32_t)m.getDistribCount(), D = (uint32_t)m.getVectSize();
    o.write((const char *)&C, 4);
    o.write((const char *)&D, 4);
    for (uint32_t c = 0; c < C; ++c) { const double w = m.weight(c); o.write((const char *)&w, 8); }
    for (uint32_t c = 0; c < C; ++c) {
        double det = 1.0;
        for (uint32_t d = 0; d < D; ++d) det *= m.getCov(c, d);
        const double cst = pow(2.0 * M_PI, -0.5 * D) / sqrt(det); // DistribGD::computeAll
        const unsigned char flag = 0;
        o.write((const char *)&cst, 8);
        o.write((const char *)&det, 8);
        o.write((const char *)&flag, 1);
        for (uint32_t d = 0; d < D; ++d) { const double v = m.getCovInv(c, d); o.write((const char *)&v, 8); }
        for (uint32_t d = 0; d < D; ++d) { const double v = m.getMean(c, d); o.write((const char *)&v, 8); }
    }
}

MatrixD readMatrixDT(const std::string &path)
{
    std::ifstream f(path.c_str());
    if (!f) throw Exception("cannot open matrix [" + path + "]");
    MatrixD m;
    if (!(f >> m.rows >> m.cols)) throw Exception("bad DT matrix header [" + path + "]");
    m.v.resize(m.rows * m.cols);
    for (double &x : m.v)
        if (!(f >> x)) throw Exception("DT matrix truncated [" + path + "]");
    return m;
}

void writeMatrixDT(const std::string &path, const MatrixD &m)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) throw Exception("cannot write [" + path + "]");
    fprintf(f, "%lu %lu\n", m.rows, m.cols);
    for (unsigned long i = 0; i < m.rows; ++i) {
        for (unsigned long j = 0; j < m.cols; ++j) fprintf(f, "%.17g ", m.v[i * m.cols + j]);
        fprintf(f, "\n");
    }
    fclose(f);
}

std::string resultLine(double llr, const std::string &clientName, const std::string &testName, const std::string &gender,
                       double threshold, bool withTimes, double start, double end)
{
    std::ostringstream o;
    o << gender << " " << clientName << " " << (llr > threshold ? 1 : 0) << " " << testName << " ";
    if (withTimes) o << start << " " << end << " ";
    o << llr;
    return o.str();
}

} // namespace liagpu
