"""ctypes loader for the CPU oracle (oracle/_build/libhk_oracle.so).  Test infrastructure only."""
import ctypes as C, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def build():
    so = os.path.join(ROOT, "oracle", "_build", "libhk_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(ROOT, "include", f) for f in ("hk.h", "hk_detmath.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.hko_lq_solve.restype = C.c_int
        L.hko_lq_solve.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp]
        L.hko_bicycle_AB.restype = None
        L.hko_bicycle_AB.argtypes = [C.c_double, dp, dp, dp]
        L.hko_cost_build.restype = None
        L.hko_cost_build.argtypes = [C.c_int, dp, dp, C.c_double, dp, dp, dp, dp, dp, dp]
        for f in ("hko_sin", "hko_cos", "hko_exp"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        L.hko_atan2.restype = C.c_double
        L.hko_atan2.argtypes = [C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def lq_solve(A, B, Q, q, R, x0, horizon=3, want_trace=False):
    A = np.ascontiguousarray(A, np.float64); N = A.shape[0]; n = 4 * N; m = 2 * N
    B = np.ascontiguousarray(B, np.float64); Q = np.ascontiguousarray(Q, np.float64)
    q = np.ascontiguousarray(q, np.float64); R = np.ascontiguousarray(R, np.float64)
    x0 = np.ascontiguousarray(x0, np.float64)
    u0 = np.zeros(2)
    tr = np.zeros((horizon + 1) * (m * n + m))
    rc = lib().hko_lq_solve(N, _p(A), _p(B), _p(Q), _p(q), _p(R), _p(x0), horizon, _p(u0), _p(tr))
    assert rc == 0, rc
    if want_trace:
        out = []
        o = 0
        for _ in range(horizon + 1):
            P = tr[o:o + m * n].reshape(m, n); o += m * n
            al = tr[o:o + m].copy(); o += m
            out.append((P.copy(), al))
        return u0, out
    return u0


def bicycle_AB(dt, initial):
    ini = np.ascontiguousarray(initial, np.float64)
    A = np.zeros(16); B = np.zeros(8)
    lib().hko_bicycle_AB(dt, _p(ini), _p(A), _p(B))
    return A.reshape(4, 4), B.reshape(4, 2)


def cost_build(target, target_w, control_w, avoid_w, opp_target, opp_w):
    avoid_w = np.ascontiguousarray(avoid_w, np.float64).reshape(2, -1)
    M = avoid_w.shape[1]; n = 4 + 4 * M
    t = np.ascontiguousarray(target, np.float64); tw = np.ascontiguousarray(target_w, np.float64)
    ot = np.ascontiguousarray(opp_target, np.float64).reshape(M, 4) if M else np.zeros((1, 4))
    ow = np.ascontiguousarray(opp_w, np.float64).reshape(M, 3) if M else np.zeros((1, 3))
    if M == 0:
        avoid_w = np.zeros((2, 1))
    Q = np.zeros(n * n); q = np.zeros(n); R = np.zeros(4)
    lib().hko_cost_build(M, _p(t), _p(tw), control_w, _p(avoid_w), _p(ot), _p(ow), _p(Q), _p(q), _p(R))
    return Q.reshape(n, n), q, R.reshape(2, 2)
