"""Host-side mirror of the reference's LQ surface: KartLQR.solveFeedbackLQR (AI/LQR/KartLQR.cs:17), batched."""
import ctypes as C
import numpy as np
from . import _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve_feedback_lqr_batch(A, B, Q, q, R, x0, horizon=3, handle=None):
    """A[b][N][4][4], B[b][N][4][2], Q[b][N][n][n], q[b][N][n], R[b][N][2][2], x0[b][n] -> u0[b][2]
    (player 0's control at t = 0, = the Vector<double> the reference returns).  Runs on the GPU through libhk."""
    A = np.ascontiguousarray(A, np.float64)
    if A.ndim != 4 or A.shape[2:] != (4, 4):
        raise ValueError("A must be [batch][N][4][4]")
    b, N = A.shape[0], A.shape[1]
    n = 4 * N
    B = np.ascontiguousarray(B, np.float64).reshape(b, N, 4, 2)
    Q = np.ascontiguousarray(Q, np.float64).reshape(b, N, n, n)
    q = np.ascontiguousarray(q, np.float64).reshape(b, N, n)
    R = np.ascontiguousarray(R, np.float64).reshape(b, N, 2, 2)
    x0 = np.ascontiguousarray(x0, np.float64).reshape(b, n)
    u0 = np.zeros((b, 2))
    L = _lib.load()
    rc = L.hk_lq_solve_batch(handle, b, N, _dp(A), _dp(B), _dp(Q), _dp(q), _dp(R), _dp(x0), int(horizon), _dp(u0))
    _lib.check(rc, handle)
    return u0


def solve_feedback_lqr(dynamics, costs, initials, horizon=3):
    """Single game with the reference's argument meaning: dynamics = [(A_i, B_i)], costs = [(Q_i, q_i, R_i)],
    initials = [x_i (4,)] -> numpy (2,)"""
    A = np.array([d[0] for d in dynamics])[None]
    B = np.array([d[1] for d in dynamics])[None]
    Q = np.array([c[0] for c in costs])[None]
    q = np.array([c[1] for c in costs])[None]
    R = np.array([c[2] for c in costs])[None]
    x0 = np.concatenate([np.asarray(x, float) for x in initials])[None]
    return solve_feedback_lqr_batch(A, B, Q, q, R, x0, horizon)[0]
