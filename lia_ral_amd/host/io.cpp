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

// ---- XML mixtures ---------------------------------------------------------------------------------------------------------
namespace {
// value of attribute `name` inside the tag text [b, e)
std::string xmlAttr(const std::string &t, size_t b, size_t e, const char *name, const std::string &path)
{
    const std::string key = std::string(name) + "=\"";
    const size_t p = t.find(key, b);
    if (p == std::string::npos || p >= e) throw Exception("XML mixture [" + path + "]: attribute " + name + " missing");
    const size_t q = t.find('"', p + key.size());
    if (q == std::string::npos || q > e) throw Exception("XML mixture [" + path + "]: unterminated attribute " + name);
    return t.substr(p + key.size(), q - p - key.size());
}
} // namespace

MixtureGD readMixtureXML(const std::string &path)
{
    const std::vector<unsigned char> raw = slurp(path);
    const std::string t(raw.begin(), raw.end());
    size_t pos = t.find("<MixtureGD");
    if (pos == std::string::npos) throw Exception("XML mixture [" + path + "]: no <MixtureGD> element");
    size_t end = t.find('>', pos);
    if (end == std::string::npos) throw Exception("XML mixture [" + path + "]: truncated header");
    const unsigned long C = strtoul(xmlAttr(t, pos, end, "distribCount", path).c_str(), nullptr, 10);
    const unsigned long D = strtoul(xmlAttr(t, pos, end, "vectSize", path).c_str(), nullptr, 10);
    if (!C || !D) throw Exception("XML mixture [" + path + "]: distribCount / vectSize must be positive");
    MixtureGD m(C, D);
    pos = end;
    for (unsigned long c = 0; c < C; ++c) {
        pos = t.find("<DistribGD", pos);
        if (pos == std::string::npos) throw Exception("XML mixture [" + path + "]: fewer <DistribGD> elements than distribCount");
        end = t.find('>', pos);
        if (strtoul(xmlAttr(t, pos, end, "i", path).c_str(), nullptr, 10) != c) throw Exception("XML mixture [" + path + "]: distributions out of order");
        m.weight(c) = strtod(xmlAttr(t, pos, end, "weight", path).c_str(), nullptr);
        const size_t close = t.find("</DistribGD>", end);
        if (close == std::string::npos) throw Exception("XML mixture [" + path + "]: unterminated <DistribGD>");
        for (int what = 0; what < 2; ++what) {
            const std::string tag = what == 0 ? "<covInv i=\"" : "<mean i=\"";
            size_t p = end;
            for (unsigned long d = 0; d < D; ++d) {
                p = t.find(tag, p);
                if (p == std::string::npos || p > close) throw Exception("XML mixture [" + path + "]: a distribution has fewer than vectSize entries");
                const unsigned long idx = strtoul(t.c_str() + p + tag.size(), nullptr, 10);
                const size_t v = t.find('>', p);
                if (idx >= D || v == std::string::npos) throw Exception("XML mixture [" + path + "]: bad entry index");
                const double val = strtod(t.c_str() + v + 1, nullptr);
                if (what == 0) m.setCovInv(c, val, idx); else m.setMean(c, val, idx);
                p = v;
            }
        }
        pos = close;
    }
    return m; // no computeAll(): the covInv bits of the file are kept
}

void writeMixtureXML(const std::string &path, const MixtureGD &m, const std::string &id)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) throw Exception("cannot write [" + path + "]");
    const unsigned long C = m.getDistribCount(), D = m.getVectSize();
    fprintf(f, "<MixtureGD version=\"1\" id=\"%s\" distribCount=\"%lu\" vectSize=\"%lu\">\n", id.c_str(), C, D);
    for (unsigned long c = 0; c < C; ++c) {
        double det = 1.0;
        for (unsigned long d = 0; d < D; ++d) det *= m.getCov(c, d);
        const double cst = pow(2.0 * M_PI, -0.5 * D) / sqrt(det); // DistribGD::computeAll
        fprintf(f, "\t<DistribGD i=\"%lu\" weight=\"%.19g\" cst=\"%.19g\" det=\"%.19g\">\n", c, m.weight(c), cst, det);
        for (unsigned long d = 0; d < D; ++d) fprintf(f, "\t\t<covInv i=\"%lu\">%.19g</covInv>\n", d, m.getCovInv(c, d));
        for (unsigned long d = 0; d < D; ++d) fprintf(f, "\t\t<mean i=\"%lu\">%.19g</mean>\n", d, m.getMean(c, d));
        fprintf(f, "\t</DistribGD>\n");
    }
    fprintf(f, "</MixtureGD>\n");
    fclose(f);
}

MixtureGD readMixture(const std::string &path)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) throw Exception("cannot open [" + path + "]");
    char c = 0;
    while (f.get(c) && (c == ' ' || c == '\n' || c == '\r' || c == '\t')) {}
    return c == '<' ? readMixtureXML(path) : readMixtureRAW(path);
}

// ---- DB matrices and per-id vector files ------------------------------------------------------------------------------------
// Header width: alize-core's Matrix<T> keeps its extents in `unsigned long` members and (to the best of what can be told
// without its sources -- alize-core is not in the LIA_RAL tree and no DB file ships with it) writes them with sizeof(member):
// 2 x 8 bytes from an LP64 build (Linux / macOS), 2 x 4 bytes from an ILP32 or LLP64 build (32-bit, Windows).  The reader
// takes whichever width makes the file size come out exactly (ambiguity is impossible: with the wide header the upper halves
// are zero, which read as a 0-column matrix under the narrow interpretation -- only an empty matrix satisfies both, and both
// readings agree on it); the writer uses this platform's `unsigned long` like an upstream build here would, or the width
// asked for (setMatrixDBHeaderBytes(4 | 8)) when a file is meant for a build of the other kind.
static int g_db_header_bytes = (int)sizeof(unsigned long);
int setMatrixDBHeaderBytes(int bytes)
{
    const int prev = g_db_header_bytes;
    if (bytes == 4 || bytes == 8) g_db_header_bytes = bytes;
    return prev;
}
MatrixD readMatrixDB(const std::string &path)
{
    const std::vector<unsigned char> b = slurp(path);
    if (b.size() < 8) throw Exception("DB matrix too short [" + path + "]");
    uint64_t r = 0, c = 0;
    size_t hdr = 0;
    if (b.size() >= 16) { // LP64 writer: two 8-byte extents
        uint64_t r8, c8;
        memcpy(&r8, b.data(), 8);
        memcpy(&c8, b.data() + 8, 8);
        if (r8 <= (1ull << 40) && c8 <= (1ull << 40) && (r8 == 0 || c8 <= (~(size_t)0 - 16) / 8 / r8) && b.size() == 16 + 8 * (size_t)r8 * (size_t)c8) { r = r8; c = c8; hdr = 16; }
    }
    if (!hdr) {           // ILP32 / LLP64 writer: two 4-byte extents
        uint32_t r4, c4;
        memcpy(&r4, b.data(), 4);
        memcpy(&c4, b.data() + 4, 4);
        if (b.size() == 8 + 8 * (size_t)r4 * c4) { r = r4; c = c4; hdr = 8; }
    }
    if (!hdr) {
        uint32_t r4, c4;
        memcpy(&r4, b.data(), 4);
        memcpy(&c4, b.data() + 4, 4);
        char msg[320];
        snprintf(msg, sizeof(msg), "DB matrix [%s]: %zu bytes fit neither an 8-byte-extent header nor a 4-byte one (%u x %u would need %zu)",
                 path.c_str(), b.size(), r4, c4, 8 + 8 * (size_t)r4 * c4);
        throw Exception(msg);
    }
    MatrixD m;
    m.rows = (unsigned long)r; m.cols = (unsigned long)c;
    m.v.resize((size_t)r * c);
    if (!m.v.empty()) memcpy(m.v.data(), b.data() + hdr, 8 * m.v.size());
    return m;
}
void writeMatrixDB(const std::string &path, const MatrixD &m)
{
    std::ofstream o(path.c_str(), std::ios::binary);
    if (!o) throw Exception("cannot write [" + path + "]");
    if (g_db_header_bytes == 8) {
        const uint64_t r = (uint64_t)m.rows, c = (uint64_t)m.cols;
        o.write((const char *)&r, 8);
        o.write((const char *)&c, 8);
    } else {
        const uint32_t r = (uint32_t)m.rows, c = (uint32_t)m.cols;
        o.write((const char *)&r, 4);
        o.write((const char *)&c, 4);
    }
    if (!m.v.empty()) o.write((const char *)m.v.data(), 8 * m.v.size());
}
MatrixD readMatrix(const std::string &path, const std::string &format)
{
    if (format == "DT") return readMatrixDT(path);
    if (format == "DB") return readMatrixDB(path);
    throw Exception("unknown matrix format [" + format + "]");
}
void writeMatrix(const std::string &path, const MatrixD &m, const std::string &format)
{
    if (format == "DT") writeMatrixDT(path, m);
    else if (format == "DB") writeMatrixDB(path, m);
    else throw Exception("unknown matrix format [" + format + "]");
}

void saveVectorsById(const std::string &dir, const std::vector<std::string> &ids, const std::string &ext, const std::vector<double> &W,
                     unsigned long rank, const std::string &format)
{
    if (W.size() != ids.size() * rank) throw Exception("saveVectorsById: W must hold one row of `rank` values per id");
    for (size_t s = 0; s < ids.size(); ++s) { // one 1 x rank matrix per id, in list order (saveWbyFile: session++)
        MatrixD y;
        y.rows = 1; y.cols = rank;
        y.v.assign(W.begin() + s * rank, W.begin() + (s + 1) * rank);
        writeMatrix(dir + ids[s] + ext, y, format);
    }
}
std::vector<double> loadVectorsById(const std::string &dir, const std::vector<std::string> &ids, const std::string &ext, unsigned long &dim,
                                    const std::string &format)
{
    dim = 0;
    std::vector<double> out;
    for (size_t k = 0; k < ids.size(); ++k) {
        const MatrixD v = readMatrix(dir + "/" + ids[k] + ext, format);
        if (k == 0) { dim = v.cols; out.assign((size_t)dim * ids.size(), 0.0); } // _vectSize = tmpVect.cols() of the first file
        if (v.rows < 1 || v.cols != dim) throw Exception("vector file [" + ids[k] + ext + "]: expected 1 x " + std::to_string(dim));
        for (unsigned long i = 0; i < dim; ++i) out[i * ids.size() + k] = v.v[i]; // _models(k, m) = tmpVect(0, k)
    }
    return out;
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
