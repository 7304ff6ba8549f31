"""The one-game-per-wave solver of 3- and 4-player games (csrc/hk_lq_mfma.h: the two dense products of the value update on the fp64
matrix core) compiled for the host — 64 threads play the lanes, the f64 MFMA is the k-ascending fma chain the hardware computes
(tools/experiments/mfma_f64_check.hip), v_readlane and the wave-level LDS ordering are barrier-guarded exchanges — against the C oracle's
solveFeedbackLQR: the controls must be identical bit for bit, on games with SolveLQR's structure and on generic dense games (random A,
B with pivoting in the m x m solve).  The same header runs on the GPU (tests/test_lq_gpu.py, the env parity tests)."""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
from oracle import lq_numpy as LQ

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lqm") / "lq_mfma_host_check")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "hierarchicalkarting_amd", "csrc"),
                           os.path.join(ROOT, "tests", "lq_mfma_host_check.cpp"), "-o", exe, "-lpthread"])
    return exe


def _dense_game(rng, N):
    """arbitrary A_i / B_i blocks and costs (what hk_lq_solve_batch accepts): the LU pivots, nothing is structurally zero"""
    n = 4 * N
    A = np.eye(4)[None] + 0.05 * rng.standard_normal((N, 4, 4))
    B = 0.1 * rng.standard_normal((N, 4, 2))
    Q = rng.standard_normal((N, n, n)) * 0.2
    Q = Q + Q.transpose(0, 2, 1) + 2.0 * np.eye(n)[None]
    q = rng.standard_normal((N, n))
    R = np.array([np.eye(2) * rng.uniform(0.1, 0.3) + 0.02 * rng.standard_normal((2, 2)) for _ in range(N)])
    x0 = rng.uniform(-20, 20, n)
    return A, B, Q, q, R, x0


@pytest.mark.parametrize("N", [3, 4])
@pytest.mark.parametrize("kind", ["bicycle", "dense"])
def test_mfma_solver_arithmetic_equals_the_oracle(harness, N, kind):
    rng = np.random.default_rng(100 * N + (kind == "dense"))
    ng = 3
    games = [LQ.random_game(rng, N) if kind == "bicycle" else _dense_game(rng, N) for _ in range(ng)]
    toks = ["%d %d" % (N, ng)]
    for gm in games:
        for arr in gm:
            toks.append(" ".join(repr(float(x)) for x in np.asarray(arr, np.float64).ravel()))
    out = subprocess.run([harness], input="\n".join(toks), capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == ng
    for g, line in enumerate(out):
        a, b, sing = line.split()
        ref = O.lq_solve(*games[g], 3)
        assert float.fromhex(a) == ref[0] and float.fromhex(b) == ref[1] and int(sing) == 0, (g, a, ref[0], b, ref[1])
