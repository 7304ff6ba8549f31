"""KartLQR.solveFeedbackLQR evaluated in MathNet's OWN arithmetic — every product of KartLQR.cs:78-117 accumulated k ascending as
`s = s + a * b` (a multiply and an add, a rounding each), `Solve` as UserLU — against the C oracle, whose contract keeps that order and
fuses the two roundings (`s = fma(a, b, s)`).  The order and the two-rounding form are what tools/mathnet_il.py reads
from the IL of the MathNet.Numerics.dll the reference ships (tests/golden/mathnet_userlu_facts.json: DoMultiplySparse, Double.Matrix's
TransposeThisAndMultiply loop, the sparse vector product, UserLU.Create / Solve).  Python floats are IEEE doubles and `s + a * b` in Python
rounds twice, so this file IS that arithmetic; the difference to the oracle is the one rounding per term the contract saves, and it is
bounded here on the golden vectors: <= 1e-13 relative on the control (measured 4e-16 .. 1.5e-15; BASELINE's tolerance: 1e-4)."""
import json, os
import numpy as np
import pytest
import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FACTS = json.load(open(os.path.join(GOLD, "mathnet_userlu_facts.json")))["facts"]


def mm(A, B):
    """C = A B, k ascending, s = s + a*b (first term: 0.0 + a*b = a*b)"""
    r, kk, c = len(A), len(B), len(B[0])
    C = [[0.0] * c for _ in range(r)]
    for i in range(r):
        for j in range(c):
            s = 0.0
            for k in range(kk):
                s = s + A[i][k] * B[k][j]
            C[i][j] = s
    return C


def tr(A):
    return [list(x) for x in zip(*A)]


def add(A, B, sign=1.0):
    return [[a + sign * b for a, b in zip(ra, rb)] for ra, rb in zip(A, B)]


def lu_solve(LHS, RHS):
    """UserLU.Create (column-oriented Doolittle, first strictly larger pivot, division by the pivot) + UserLU.Solve, as the facts state"""
    m = len(LHS); nb = len(RHS[0])
    LU = [row[:] for row in LHS]; piv = list(range(m))
    for j in range(m):
        col = [LU[i][j] for i in range(m)]
        for i in range(m):
            kmax = min(i, j); s = 0.0
            for k in range(kmax):
                s = s + LU[i][k] * col[k]
            col[i] -= s; LU[i][j] = col[i]
        p = j
        for i in range(j + 1, m):
            if abs(col[i]) > abs(col[p]):
                p = i
        if p != j:
            LU[p], LU[j] = LU[j], LU[p]; piv[j] = p
        if j < m and LU[j][j] != 0.0:
            for i in range(j + 1, m):
                LU[i][j] /= LU[j][j]
    B = [row[:] for row in RHS]
    for i in range(m):
        if piv[i] != i:
            B[i], B[piv[i]] = B[piv[i]], B[i]
    for k in range(m):
        for i in range(k + 1, m):
            for j in range(nb):
                B[i][j] = B[i][j] - B[k][j] * LU[i][k]
    for k in range(m - 1, -1, -1):
        for j in range(nb):
            B[k][j] /= LU[k][k]
        for i in range(k):
            for j in range(nb):
                B[i][j] = B[i][j] - B[k][j] * LU[i][k]
    return B


def solve_mathnet(As, Bl, Qs, qs, Rs, x0, horizon):
    """KartLQR.cs:17-128, line by line, on lists of Python floats"""
    N = len(As); n = 4 * N; m = 2 * N
    A = [[0.0] * n for _ in range(n)]
    Bs = []
    for i in range(N):
        for r in range(4):
            for c in range(4):
                A[4 * i + r][4 * i + c] = float(As[i][r][c])
        B = [[0.0] * 2 for _ in range(n)]
        for r in range(4):
            for c in range(2):
                B[4 * i + r][c] = float(Bl[i][r][c])
        Bs.append(B)
    Zs = [[[float(v) for v in row] for row in Q] for Q in Qs]
    etas = [[[float(v)] for v in q] for q in qs]
    Rm = [[[float(v) for v in row] for row in R] for R in Rs]
    P = alpha = None
    for t in range(horizon, -1, -1):
        LHS = [[0.0] * m for _ in range(m)]
        for i in range(N):                       # :68-88: column block i is the stack over j of Bs[i]' (Zs[i] Bs[j]) (+ R on the diagonal)
            for j in range(N):
                blk = mm(tr(Bs[i]), mm(Zs[i], Bs[j]))
                if i == j:
                    blk = add(Rm[i], blk)
                for r in range(2):
                    for c in range(2):
                        LHS[2 * j + r][2 * i + c] = blk[r][c]
        RHSm = []; RHSv = []
        for i in range(N):
            RHSm += mm(tr(Bs[i]), mm(Zs[i], A))
            RHSv += mm(tr(Bs[i]), etas[i])
        P = lu_solve(LHS, RHSm); alpha = lu_solve(LHS, RHSv)
        F = [[0.0] * n for _ in range(n)]; beta = [[0.0] for _ in range(n)]
        for k in range(N):                       # :107-108: Aggregate from a zero matrix / vector
            F = add(F, mm(Bs[k], P[2 * k:2 * k + 2]))
            beta = add(beta, mm(Bs[k], alpha[2 * k:2 * k + 2]), -1.0)
        F = add(A, F, -1.0)
        nZ, ne = [], []
        for i in range(N):
            Pi = P[2 * i:2 * i + 2]; ai = alpha[2 * i:2 * i + 2]
            Zi = add(add(Qs_f[i], mm(tr(Pi), mm(Rm[i], Pi))), mm(tr(F), mm(Zs[i], F)))
            nZ.append(Zi)                        # (:113-114: etas[i] is computed AFTER Zs[i] was overwritten, i.e. with the new one)
            ne.append(add(add(qs_f[i], mm(tr(Pi), mm(Rm[i], ai))), mm(tr(F), add(etas[i], mm(Zi, beta)))))
        Zs, etas = nZ, ne
    P0 = P[0:2]; a0 = alpha[0:2]
    x = [[float(v)] for v in x0]
    u = add([[-v for v in row] for row in mm(P0, x)], a0, -1.0)      # -P * initial - alpha  (unary minus, then the products, then the subtraction)
    return [u[0][0], u[1][0]]


@pytest.mark.parametrize("N", [1, 2, 3, 4])
def test_fused_contract_differs_from_mathnets_two_roundings_by_rounding_only(N):
    global Qs_f, qs_f
    cases = json.load(open(os.path.join(GOLD, "lq_N%d.json" % N)))["cases"]
    worst = 0.0
    for c in cases[:12]:
        Qs_f = [[[float(v) for v in row] for row in Q] for Q in c["Q"]]
        qs_f = [[[float(v)] for v in q] for q in c["q"]]
        u_m = solve_mathnet(c["A"], c["B"], c["Q"], c["q"], c["R"], c["x0"], c["horizon"])
        u_o, _ = O.lq_solve(c["A"], c["B"], c["Q"], c["q"], c["R"], c["x0"], c["horizon"], want_trace=True)
        scale = max(1.0, abs(u_o[0]), abs(u_o[1]))
        worst = max(worst, abs(u_m[0] - u_o[0]) / scale, abs(u_m[1] - u_o[1]) / scale)
    assert worst <= 1e-13, worst


def test_the_order_used_above_is_the_one_read_from_the_il():
    need = ["DoMultiplySparse: rows of this, its stored entries k ascending",
            "TransposeThisAndMultiply(Matrix) on a SparseMatrix runs Double.Matrix's loop",
            "SparseMatrix.DoTransposeThisAndMultiply exists for (Vector, Vector) only",
            "SparseMatrix.DoMultiply(Matrix, Matrix) hands sparse operands to DoMultiplySparse"]
    for n in need:
        assert any(k.startswith(n) and v is True for k, v in FACTS.items()), n
