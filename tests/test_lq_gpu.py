"""hk_lq_solve_batch (HIP, gfx950) vs the CPU oracle: bit-exact (the arithmetic contract of hk_lq_core.h), and vs the
committed golden vectors from the independent numpy mirror within 1e-9."""
import json, os
import numpy as np
import pytest
import oracle_lib as O
from oracle import lq_numpy as LQ

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _stack(cases):
    return [np.array([c[k] for c in cases]) for k in ("A", "B", "Q", "q", "R", "x0")]


@pytest.mark.parametrize("N", [1, 2, 3, 4, 8])
def test_golden(N):
    import hierarchicalkarting_amd as hk
    with open(os.path.join(GOLD, "lq_N%d.json" % N)) as f:
        cases = json.load(f)["cases"]
    A, B, Q, q, R, x0 = _stack(cases)
    u = hk.solve_feedback_lqr_batch(A, B, Q, q, R, x0, 3)
    for i, c in enumerate(cases):
        scale = max(1.0, np.abs(c["u0"]).max())
        assert np.abs(u[i] - np.array(c["u0"])).max() <= 1e-9 * scale
        uo = O.lq_solve(c["A"], c["B"], c["Q"], c["q"], c["R"], c["x0"], 3)
        assert np.array_equal(u[i], uo), (u[i], uo)          # bit-exact vs the oracle


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_batch_bit_exact(N):
    import hierarchicalkarting_amd as hk
    rng = np.random.default_rng(50 + N)
    games = [LQ.random_game(rng, N) for _ in range(203 if N <= 4 else 37)]   # ragged vs the games-per-wave packing
    A = np.array([g[0] for g in games]); B = np.array([g[1] for g in games]); Q = np.array([g[2] for g in games])
    q = np.array([g[3] for g in games]); R = np.array([g[4] for g in games]); x0 = np.array([g[5] for g in games])
    u = hk.solve_feedback_lqr_batch(A, B, Q, q, R, x0, 3)
    for i in range(len(games)):
        uo = O.lq_solve(A[i], B[i], Q[i], q[i], R[i], x0[i], 3)
        assert np.array_equal(u[i], uo), (i, u[i], uo)


def test_generic_dense_inputs_and_pivoting():
    """solveFeedbackLQR is generic in A_i/B_i/Q_i/R_i: dense random blocks, dense R, and an LHS that forces row
    pivoting (tiny R, large off-diagonal coupling)."""
    import hierarchicalkarting_amd as hk
    rng = np.random.default_rng(9)
    for N in (2, 4):
        n = 4 * N
        A = rng.normal(size=(64, N, 4, 4)) * 0.3 + np.eye(4)
        B = rng.normal(size=(64, N, 4, 2)) * 0.5
        Q = rng.normal(size=(64, N, n, n)); Q = Q + Q.transpose(0, 1, 3, 2)
        q = rng.normal(size=(64, N, n))
        R = rng.normal(size=(64, N, 2, 2)) * 1e-3
        x0 = rng.normal(size=(64, n))
        u = hk.solve_feedback_lqr_batch(A, B, Q, q, R, x0, 3)
        for i in range(64):
            uo = O.lq_solve(A[i], B[i], Q[i], q[i], R[i], x0[i], 3)
            assert np.array_equal(u[i], uo), (N, i, u[i], uo)


def test_horizon_and_edge_cases():
    import hierarchicalkarting_amd as hk
    rng = np.random.default_rng(3)
    g = LQ.random_game(rng, 3)
    for hz in (0, 1, 5):
        u = hk.solve_feedback_lqr_batch(*[np.array(x)[None] for x in g], hz)
        assert np.array_equal(u[0], O.lq_solve(*g, hz))
    # empty batch
    z = hk.solve_feedback_lqr_batch(np.zeros((0, 2, 4, 4)), np.zeros((0, 2, 4, 2)), np.zeros((0, 2, 8, 8)),
                                    np.zeros((0, 2, 8)), np.zeros((0, 2, 2, 2)), np.zeros((0, 8)))
    assert z.shape == (0, 2)
    # unsupported / invalid
    from hierarchicalkarting_amd import _lib
    with pytest.raises(_lib.HkError) as e:
        hk.solve_feedback_lqr_batch(np.zeros((1, 9, 4, 4)), np.zeros((1, 9, 4, 2)), np.zeros((1, 9, 36, 36)),
                                    np.zeros((1, 9, 36)), np.zeros((1, 9, 2, 2)), np.zeros((1, 36)))      # > 8 players
    assert e.value.code == _lib.HK_ERR_UNSUPPORTED
    # singular LHS (all-zero costs and R) reports instead of returning garbage
    with pytest.raises(_lib.HkError) as e:
        hk.solve_feedback_lqr_batch(np.tile(np.eye(4), (1, 2, 1, 1)), np.zeros((1, 2, 4, 2)), np.zeros((1, 2, 8, 8)),
                                    np.zeros((1, 2, 8)), np.zeros((1, 2, 2, 2)), np.zeros((1, 8)))
    assert e.value.code == _lib.HK_ERR_SINGULAR
