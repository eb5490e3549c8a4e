#!/usr/bin/env python3
"""Golden vectors for the THIRD-PARTY numerics the reference calls and that ARE in the container: Eigen 3.1.2, vendored under
/root/reference/include/Eigen.  Run in the build container only (the reference tree does not travel):

    python tests/golden/make_eigen_fixtures.py          ->  tests/golden/eigen_conventions.npz

A small driver -- written here, not taken from the reference -- is compiled against those headers and calls Eigen the way the cited
lines do; the outputs pin the CONVENTIONS our restatements have to match (ordering, normalisation, which triangle, log-det from
the Cholesky diagonal), not the LIA_RAL loops around them:
  * Eigen::EigenSolver on a symmetric matrix, as PldaDev::computeEigenProblem (PldaTools.cpp:1490-1535: eigenvalues sorted
    descending, eigenvector columns REORDERED with them) and as TVAcc::computeEigenProblem (AccumulateTVStat.cpp:3056-3102: eigenvalues
    sorted descending, but eigenVect(k, j) = real(V(k, j)) -- the FIRST `rank` columns in Eigen's own order, not reordered);
  * EigenSolver on the non-symmetric W^-1 B of PldaDev::computeLDA (PldaTools.cpp:1381-1414);
  * (n FTJF + I).inverse() and alpha_n = 2 sum log diag(K_n.llt().matrixL()) of PldaTest::pldaScoring (PldaTools.cpp:4226-4249);
  * c.llt().matrixL().transpose() and X * E.inverse() of PldaModel::mStep (PldaTools.cpp:2795, :2808).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EIGEN = "/root/reference/include"

DRIVER = r'''
#include <Eigen/Dense>
#include <Eigen/Eigenvalues>
#include <cstdio>
#include <vector>
static Eigen::MatrixXd readm(FILE *f) {
    int r, c; if (fscanf(f, "%d %d", &r, &c) != 2) throw 1;
    Eigen::MatrixXd m(r, c);
    for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) { double v; if (fscanf(f, "%lf", &v) != 1) throw 1; m(i, j) = v; }
    return m;
}
static void writem(const char *name, const Eigen::MatrixXd &m) {
    printf("%s %d %d\n", name, (int)m.rows(), (int)m.cols());
    for (int i = 0; i < m.rows(); ++i) { for (int j = 0; j < m.cols(); ++j) printf("%.17g ", m(i, j)); printf("\n"); }
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "r");
    Eigen::MatrixXd S = readm(f), EP = readm(f), FTJF = readm(f), Cm = readm(f), Xh = readm(f);
    fclose(f);
    {   // symmetric input
        Eigen::EigenSolver<Eigen::MatrixXd> es(S);
        Eigen::MatrixXd val(S.rows(), 2), V(S.rows(), S.cols()), Vi(S.rows(), S.cols());
        for (int i = 0; i < S.rows(); ++i) { val(i, 0) = es.eigenvalues()[i].real(); val(i, 1) = es.eigenvalues()[i].imag(); }
        Eigen::MatrixXcd Vc = es.eigenvectors();
        for (int i = 0; i < S.rows(); ++i) for (int j = 0; j < S.cols(); ++j) { V(i, j) = Vc(i, j).real(); Vi(i, j) = Vc(i, j).imag(); }
        writem("sym_val", val); writem("sym_vec", V); writem("sym_vec_imag", Vi);
    }
    {   // W^-1 B
        Eigen::EigenSolver<Eigen::MatrixXd> es(EP);
        Eigen::MatrixXd val(EP.rows(), 2), V(EP.rows(), EP.cols()), Vi(EP.rows(), EP.cols());
        for (int i = 0; i < EP.rows(); ++i) { val(i, 0) = es.eigenvalues()[i].real(); val(i, 1) = es.eigenvalues()[i].imag(); }
        Eigen::MatrixXcd Vc = es.eigenvectors();
        for (int i = 0; i < EP.rows(); ++i) for (int j = 0; j < EP.cols(); ++j) { V(i, j) = Vc(i, j).real(); Vi(i, j) = Vc(i, j).imag(); }
        writem("lda_val", val); writem("lda_vec", V); writem("lda_vec_imag", Vi);
    }
    for (int n = 1; n <= 3; ++n) {
        Eigen::MatrixXd tmpK = n * FTJF + Eigen::MatrixXd::Identity(FTJF.rows(), FTJF.rows());
        Eigen::MatrixXd K = tmpK.inverse();
        Eigen::MatrixXd a = K.llt().matrixL();
        double alpha = 0.0;
        for (int i = 0; i < a.rows(); ++i) alpha += log(a(i, i));
        alpha *= 2.0;
        char nm[32];
        snprintf(nm, sizeof nm, "K_%d", n); writem(nm, K);
        Eigen::MatrixXd al(1, 1); al(0, 0) = alpha;
        snprintf(nm, sizeof nm, "alpha_%d", n); writem(nm, al);
    }
    {
        Eigen::MatrixXd R = Cm.llt().matrixL().transpose();
        writem("chol_upper", R);
        Eigen::MatrixXd FG = Xh * Cm.inverse();
        writem("x_times_inverse", FG);
    }
    return 0;
}
'''


def main():
    if not os.path.isdir(os.path.join(EIGEN, "Eigen")):
        sys.exit("the vendored Eigen of the reference is not here (%s): run this in the build container" % EIGEN)
    rng = np.random.default_rng(20260928)
    n = 12
    A = rng.normal(size=(n, 3 * n)); S = A @ A.T / (3 * n) + 0.05 * np.diag(rng.uniform(0.5, 2.0, n))      # a covariance (EFR: Sigma)
    Wm = rng.normal(size=(n, 4 * n)); W = Wm @ Wm.T / (4 * n)                                                # within-class, SPD
    Bm = rng.normal(size=(n, 5)); B = Bm @ Bm.T / 5                                                           # between-class, rank 5
    EP = np.linalg.inv(W) @ B                                                                                 # only an INPUT of the pinned call
    r = 8
    Fm = rng.normal(size=(r, 3 * r)); FTJF = Fm @ Fm.T / r
    Cq = rng.normal(size=(r, 4 * r)); Cm = Cq @ Cq.T / (4 * r)
    Xh = rng.normal(size=(n, r))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "eigen_driver.cpp"); exe = os.path.join(td, "eigen_driver"); inp = os.path.join(td, "in.txt")
        open(src, "w").write(DRIVER)
        with open(inp, "w") as f:
            for m in (S, EP, FTJF, Cm, Xh):
                f.write("%d %d\n" % m.shape)
                for row in m:
                    f.write(" ".join("%.17g" % v for v in row) + "\n")
        subprocess.check_call(["g++", "-O1", "-w", "-I", EIGEN, "-o", exe, src])
        out = subprocess.check_output([exe, inp], text=True).split("\n")
    res, i = {}, 0
    while i < len(out):
        if not out[i].strip():
            i += 1; continue
        name, rr, cc = out[i].split(); rr, cc = int(rr), int(cc)
        res[name] = np.array([[float(v) for v in out[i + 1 + k].split()] for k in range(rr)]).reshape(rr, cc)
        i += 1 + rr
    res.update(S=S, W=W, B=B, EP=EP, FTJF=FTJF, Cm=Cm, Xh=Xh, eigen_version=np.array("3.1.2"))
    np.savez_compressed(os.path.join(HERE, "eigen_conventions.npz"), **res)
    print("wrote eigen_conventions.npz:", sorted(res))


if __name__ == "__main__":
    main()
