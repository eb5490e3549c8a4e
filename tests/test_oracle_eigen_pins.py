"""The oracle's linear-algebra conventions against the third-party numerics the reference calls and that ARE in the build
container: Eigen 3.1.2 (vendored under the reference's include/Eigen).  tests/golden/eigen_conventions.npz was written by
tests/golden/make_eigen_fixtures.py (a driver of ours, compiled against those headers, calling Eigen as the cited lines do).
This pins CONVENTIONS -- ordering, normalisation, triangle, log-determinant from the Cholesky diagonal -- not the LIA_RAL loops
around them; rows 11-19 of SURVEY.md 8(a) stay "parity unpinned" for everything else.

What cannot be pinned: the SIGN of an eigenvector.  Eigen::EigenSolver returns unit-norm vectors whose sign follows its QR
iteration; no rule reproduces it from the matrix.  A sign flip of an eigenvector flips one ROW of the EFR / sphNorm / LDA matrix,
i.e. one coordinate of every normalised vector alike: cosine, Mahalanobis, two-covariance and PLDA scores do not change."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigen_conventions.npz")


@pytest.fixture(scope="module")
def k():
    return np.load(GOLD)


def _up_to_sign(a, b):
    """max |a_j - s_j b_j| over columns with s_j = +-1."""
    s = np.sign((a * b).sum(0))
    return float(np.max(np.abs(a - b * s)))


def test_sym_eigen_matches_eigensolver_in_pldadev_order(k):
    """PldaDev::computeEigenProblem (PldaTools.cpp:1490-1535): eigenvalues descending, eigenVect(k, j) = real(V(k, EV[j].idx))."""
    S = k["S"]
    assert np.all(k["sym_val"][:, 1] == 0.0) and np.all(k["sym_vec_imag"] == 0.0)          # a symmetric matrix: real spectrum
    native = k["sym_val"][:, 0]
    assert not np.all(np.diff(native) <= 0)                  # EigenSolver's own order is NOT sorted: the reordering matters
    order = np.argsort(-native, kind="stable")               # LKVector::descendingSort
    vect, val = orc.sym_eigen(S)
    assert np.max(np.abs(val - native[order])) < 1e-12
    assert np.allclose(np.linalg.norm(k["sym_vec"], axis=0), 1.0, atol=1e-14)       # unit norm, like orc_sym_eigen's Jacobi vectors
    assert _up_to_sign(vect, k["sym_vec"][:, order]) < 1e-9
    r = 5
    vr, valr = orc.sym_eigen(S, rank=r)                      # `rank` leading pairs
    assert np.array_equal(valr, val[:r]) and np.array_equal(vr, vect[:, :r])
    # EFR / sphNorm matrix (:1866-1900): (V diag(1 / sqrt(lambda)))^T -- rows equal up to sign
    M_ref = (k["sym_vec"][:, order] / np.sqrt(native[order])).T
    M = orc.dev_efr_matrix(S)
    assert _up_to_sign(M.T, M_ref.T) < 1e-9
    assert np.max(np.abs(M @ S @ M.T - np.eye(S.shape[0]))) < 1e-10                  # it whitens S, whatever the signs


def test_tvacc_eigen_branch_keeps_eigens_own_vector_order(k):
    """TVAcc::computeEigenProblem, Eigen branch (AccumulateTVStat.cpp:3056-3102): eigenVal is sorted, eigenVect(k, j) = real(V(k, j))
    is NOT -- the first `rank` columns in EigenSolver's order (the LAPACK branch, :3032-3036, reorders them).  With rank = rankT,
    as IvExtractor calls it, that is the same basis in another column order, and estimateWEigenDecomposition does not depend on the
    order: D is rebuilt from Q by approximateTcTc (column i of D belongs to column i of Q).  Shown here on the oracle."""
    rng = np.random.default_rng(7)
    C, D, R, U = 6, 4, 5, 3
    Tm = rng.normal(0, 0.3, (R, C * D)); w = rng.dirichlet(np.ones(C))
    N = rng.gamma(1.0, 2.0, (U, C)); F = rng.normal(size=(U, C * D))
    Wc = orc.tv_weighted_cov(Tm, w)
    Q, _ = orc.sym_eigen(Wc)                                 # sorted order
    perm = np.array([3, 0, 4, 1, 2])                         # some other order, e.g. the solver's own
    Wa = orc.tv_estimate_w_eigen(N, F, Tm, orc.tv_approximate_tctc(Tm, Q, C), Q)
    Wb = orc.tv_estimate_w_eigen(N, F, Tm, orc.tv_approximate_tctc(Tm, Q[:, perm], C), Q[:, perm])
    assert np.max(np.abs(Wa - Wb)) < 1e-12 * max(1.0, np.max(np.abs(Wa)))
    sg = np.array([1, -1, -1, 1, -1.0])                      # ... nor on the signs
    Wc_ = orc.tv_estimate_w_eigen(N, F, Tm, orc.tv_approximate_tctc(Tm, Q * sg, C), Q * sg)
    assert np.max(np.abs(Wa - Wc_)) < 1e-12 * max(1.0, np.max(np.abs(Wa)))
    assert not np.all(np.diff(k["sym_val"][:, 0]) <= 0)      # and Eigen's order really differs from the sorted one


def test_lda_matches_eigensolver_on_winv_b(k):
    """PldaDev::computeLDA (PldaTools.cpp:1381-1414): the ldaRank leading eigenvectors of W^-1 B (non-symmetric EigenSolver, unit
    norm) as the rows of ldaMat.  orc_dev_lda solves the symmetric form L^-1 B L^-T instead: same eigenvalues, same directions."""
    W, B = k["W"], k["B"]
    rank = 5                                                 # B has rank 5
    val = k["lda_val"][:, 0]
    order = np.argsort(-val, kind="stable")[:rank]
    assert np.max(np.abs(k["lda_val"][order, 1])) == 0.0 and np.max(np.abs(k["lda_vec_imag"][:, order])) == 0.0
    lda, ev = orc.dev_lda(W, B, rank)
    assert np.max(np.abs(ev - val[order]) / val[order]) < 1e-10
    ref = k["lda_vec"][:, order]
    assert np.allclose(np.linalg.norm(ref, axis=0), 1.0, atol=1e-13)
    assert _up_to_sign(lda.T, ref) < 1e-8


def test_plda_k_inverse_and_logdet_conventions(k):
    """PldaTest::pldaScoring (PldaTools.cpp:4226-4249): K_n = (n FTJF + I).inverse(), alpha_n = 2 sum log diag(K_n.llt().matrixL())
    = log det K_n.  orc_invert (Gauss-Jordan) and the Cholesky diagonal of orc_upper_cholesky give the same numbers."""
    FTJF = k["FTJF"]
    r = FTJF.shape[0]
    for n in (1, 2, 3):
        K = orc.invert(n * FTJF + np.eye(r))
        assert np.max(np.abs(K - k["K_%d" % n])) < 1e-13
        U = orc.upper_cholesky(K)
        alpha = 2.0 * np.sum(np.log(np.diag(U)))
        assert abs(alpha - float(k["alpha_%d" % n][0, 0])) < 1e-12
        assert abs(alpha - np.linalg.slogdet(K)[1]) < 1e-12


def test_cholesky_triangle_and_right_inverse_conventions(k):
    """PldaModel::mStep (PldaTools.cpp:2795, :2808): c.llt().matrixL().transpose() is the UPPER factor R with R^T R = c -- what
    DoubleSquareMatrix::upperCholesky (orc_upper_cholesky) returns; X * E.inverse() through orc_invert."""
    Cm = k["Cm"]
    U = orc.upper_cholesky(Cm)
    assert np.max(np.abs(U - k["chol_upper"])) < 1e-13
    assert np.max(np.abs(np.tril(U, -1))) == 0.0 and np.max(np.abs(U.T @ U - Cm)) < 1e-13
    assert np.max(np.abs(k["Xh"] @ orc.invert(Cm) - k["x_times_inverse"])) < 1e-11
