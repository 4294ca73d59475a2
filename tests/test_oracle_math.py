"""Pins of the oracle's math floor (CPU).

The reference's only tests on this path are the vendored Sophus property tests
(thirdparty/Sophus/sophus/tests.hpp:43-201, vectors of thirdparty/Sophus/sophus/test_se3.cpp:38-92, commented
out of its build).  They are restated here against oracle/orc_math.h with the same tolerances
(SMALL_EPS = 1e-10 for double; adjoint 20x, expMap 10x).
"""
import numpy as np
import pytest
import scipy.linalg

from oracle import oracle as orc

SMALL_EPS = 1e-10


def _p(a):
    return a.ctypes.data_as(orc.C.c_void_p)


def se3_exp(t):
    t = np.ascontiguousarray(t, dtype=np.float64)
    o = np.zeros(12)
    orc.lib().orc_se3_exp12(_p(t), _p(o))
    return o


def se3_log(T):
    o = np.zeros(6)
    orc.lib().orc_se3_log12(_p(np.ascontiguousarray(T)), _p(o))
    return o


def se3_mul(A, B):
    o = np.zeros(12)
    orc.lib().orc_se3_mul12(_p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(B)), _p(o))
    return o


def se3_inv(A):
    o = np.zeros(12)
    orc.lib().orc_se3_inv12(_p(np.ascontiguousarray(A)), _p(o))
    return o


def se3_adj(A):
    o = np.zeros(36)
    orc.lib().orc_se3_adj12(_p(np.ascontiguousarray(A)), _p(o))
    return o.reshape(6, 6)


def mat4(T):
    M = np.eye(4)
    M[:3, :3] = T[:9].reshape(3, 3)
    M[:3, 3] = T[9:]
    return M


def hat(x):
    u, w = x[:3], x[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def vee(M):
    return np.array([M[0, 3], M[1, 3], M[2, 3], M[2, 1], M[0, 2], M[1, 0]])


def so3(w, t):
    return se3_mul(np.concatenate([np.eye(3).reshape(-1), np.asarray(t, float)]),
                   se3_exp(np.concatenate([[0, 0, 0], np.asarray(w, float)])))


def group_elements():
    """se3_vec of test_se3.cpp:40-62 (SE3Type(SO3::exp(w), t) = translation t, rotation exp(w))."""
    pi = np.pi
    v = [so3([0.2, 0.5, 0.0], [0, 0, 0]), so3([0.2, 0.5, -1.0], [10, 0, 0]), so3([0, 0, 0], [0, 100, 5]),
         so3([0, 0, 0.00001], [0, 0, 0]), so3([0, 0, 0.00001], [0, -0.00000001, 0.0000000001]),
         so3([0, 0, 0.00001], [0.01, 0, 0]), so3([pi, 0, 0], [4, -5, 0])]
    v.append(se3_mul(se3_mul(so3([0.2, 0.5, 0.0], [0, 0, 0]), so3([pi, 0, 0], [0, 0, 0])),
                     so3([-0.2, -0.5, -0.0], [0, 0, 0])))
    v.append(se3_mul(se3_mul(so3([0.3, 0.5, 0.1], [2, 0, -7]), so3([pi, 0, 0], [0, 0, 0])),
                     so3([-0.3, -0.5, -0.1], [0, 6, 0])))
    return v


TANGENTS = [np.array(t, dtype=np.float64) for t in
            ([0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1],
             [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0])]


def test_adjoint():  # tests.hpp:43-68
    for T in group_elements():
        Ad, M, Mi = se3_adj(T), mat4(T), mat4(se3_inv(T))
        for x in TANGENTS:
            ad1 = Ad @ x
            ad2 = vee(M @ hat(x) @ Mi)
            assert np.linalg.norm(ad1 - ad2) <= 20 * SMALL_EPS


def test_exp_log():  # tests.hpp:70-88
    for i, T in enumerate(group_elements()):
        T2 = se3_exp(se3_log(T))
        # rotations by exactly pi are on the cut of the atan-based log: allow the looser bound there
        tol = 1e-7 if i >= 6 else SMALL_EPS
        assert np.linalg.norm(mat4(T) - mat4(T2)) <= tol, i


def test_exp_map():  # tests.hpp:90-112
    for x in TANGENTS:
        assert np.linalg.norm(mat4(se3_exp(x)) - scipy.linalg.expm(hat(x))) <= 10 * SMALL_EPS * max(1, np.linalg.norm(x))


def test_group_action_and_inverse():  # tests.hpp:114-133
    p = np.array([1.0, 2.0, 4.0])
    for T in group_elements():
        M = mat4(T)
        assert np.allclose(T[:9].reshape(3, 3) @ p + T[9:], (M @ np.append(p, 1))[:3], atol=SMALL_EPS)
        assert np.linalg.norm(mat4(se3_mul(T, se3_inv(T))) - np.eye(4)) <= 1e-9


def test_ldlt_solve_residual():
    """Stand-in for Eigen `.ldlt().solve` (OB/EnergyFunctional.cpp:1148): pinned by residual norm."""
    rng = np.random.default_rng(3)
    for n in (8, 28, 100):
        B = rng.normal(size=(n, n))
        A = B @ B.T + np.diag(rng.uniform(1, 1e6, n))
        b = rng.normal(size=n)
        x = np.zeros(n)
        assert orc.lib().orc_solve_ldlt(_p(np.ascontiguousarray(A)), _p(b), _p(x), n) == 0
        assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b) * np.linalg.cond(A) ** 0.5
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-8, atol=1e-12)
    # semi-definite: a zero pivot contributes nothing (Eigen semantics)
    A = np.diag([2.0, 0.0, 5.0])
    b = np.array([2.0, 0.0, 10.0])
    x = np.zeros(3)
    orc.lib().orc_solve_ldlt(_p(A), _p(b), _p(x), 3)
    assert np.allclose(x, [1, 0, 2])


def test_pyr_levels():  # util/globalCalib.cpp:39-52 on the reference's dataset sizes
    assert orc.pyr_levels(752, 480) == 5   # EuRoC
    assert orc.pyr_levels(512, 512) == 4   # TUM-VI
    assert orc.pyr_levels(1232, 368) == 5  # KITTI 00
    assert orc.pyr_levels(1216, 368) == 5  # KITTI 04-12 crop
    assert orc.pyr_levels(96, 64) == 2 and orc.pyr_levels(160, 128) == 3
