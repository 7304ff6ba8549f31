"""KartLQR.cs:104-105 `LHS.Solve(...)` is MathNet.Numerics' UserLU; the oracle's lu_solve (oracle/hk_oracle_lq.c) and the kernels
(csrc/hk_lq_core.h) restate it.  tools/mathnet_il.py reads the IL of the binary the reference ships and checks the pivot rule, the order
of every accumulation and the substitution order; the facts and the binary's SHA-256 are committed (tests/golden/mathnet_userlu_facts.json).
Where the reference is present the facts are derived again from the binary."""
import hashlib
import json
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mathnet_il as M

EXPECTED = {
    "Double.Matrix.LU() calls UserLU.Create",
    "SparseMatrix does not override LU() (DenseMatrix does)",
    "Create: kmax = Math.Min(i, j), s = 0.0",
    "Create: s = s + LU.At(i, k) * col[k]  (mul, then add; k ascending)",
    "Create: col[i] -= s, then stored to LU[i, j]",
    "Create: pivot = first row with STRICTLY larger |col[i]| (ble.un.s skips on <= or NaN)",
    "Create: if LU[j, j] != 0.0 the entries below are DIVIDED by it",
}


def test_committed_facts_cover_what_the_restatement_relies_on():
    facts = json.load(open(M.FACTS))
    assert EXPECTED <= set(k for k, v in facts["facts"].items() if v is True)
    assert facts["facts"]["Matrix<T>.Solve overloads factor with LU() when square"] >= 2
    assert sum(k.startswith("Solve(") and v is True for k, v in facts["facts"].items()) == 2       # matrix and vector right-hand sides


def test_the_oracle_states_the_same_algorithm():
    """the restatement in the oracle, read as text: strict '>' pivot test on fabs, mul-then-add accumulation (the oracle is built with
    -ffp-contract=off), division by the pivot, forward then backward substitution with the division first"""
    src = open(os.path.join(ROOT, "oracle", "hk_oracle_lq.c")).read()
    body = src[src.index("static int lu_solve("):src.index("/* KartLQR.solveFeedbackLQR")]
    assert "if (fabs(col[i]) > fabs(col[p])) p = i;" in body
    assert "int kmax = i < j ? i : j;" in body and "s += LU[i * m + k] * col[k];" in body
    assert "LU[i * m + j] /= LU[j * m + j];" in body
    assert body.index("for (int k = 0; k < m; k++)\n        for (int i = k + 1; i < m; i++)") < body.index("for (int k = m - 1; k >= 0; k--)")
    assert "Bm[k * nb + j] /= LU[k * m + k];" in body
    assert "-ffp-contract=off" in open(os.path.join(ROOT, "oracle", "Makefile")).read()


@pytest.mark.skipif(not os.path.exists(M.DLL), reason="the reference (and its MathNet.Numerics.dll) is not on this machine")
def test_facts_rederived_from_the_reference_binary():
    facts = json.load(open(M.FACTS))
    asm = M.Assembly(M.DLL)
    assert hashlib.sha256(asm.d).hexdigest() == facts["sha256"]
    assert M.check(asm) == facts["facts"]
