"""The C oracle of KartLQR.solveFeedbackLQR / costs / dynamics vs (i) the committed golden vectors emitted by the
independent numpy mirror and (ii) the numpy mirror run live.  Tolerance 1e-9 relative on controls (target of BASELINE.md)."""
import json, os
import numpy as np
import pytest
import oracle_lib as O
from oracle import lq_numpy as LQ

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(N):
    with open(os.path.join(GOLD, "lq_N%d.json" % N)) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("N", [1, 2, 3, 4, 8])
def test_oracle_matches_golden(N):
    for c in load(N):
        u0, tr = O.lq_solve(c["A"], c["B"], c["Q"], c["q"], c["R"], c["x0"], c["horizon"], want_trace=True)
        scale = max(1.0, np.abs(c["u0"]).max())
        assert np.abs(u0 - np.array(c["u0"])).max() <= 1e-9 * scale
        P_last, a_last = tr[-1]
        assert np.allclose(P_last, np.array(c["P_last"]), rtol=1e-9, atol=1e-10)
        assert np.allclose(a_last, np.array(c["alpha_last"]), rtol=1e-9, atol=1e-9)
        P0, a0 = tr[0]
        assert np.allclose(P0, np.array(c["P_first"]), rtol=1e-10, atol=1e-11)


def test_numpy_mirror_reproduces_golden():
    for N in (2, 4):
        for c in load(N):
            u0 = LQ.solve_feedback_lqr(np.array(c["A"]), np.array(c["B"]), np.array(c["Q"]), np.array(c["q"]),
                                       np.array(c["R"]), np.array(c["x0"]), c["horizon"])
            assert np.allclose(u0, c["u0"], rtol=1e-12, atol=1e-12)


def test_dynamics_and_costs_agree():
    rng = np.random.default_rng(7)
    for _ in range(50):
        ini = [rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 15), rng.uniform(0, 6.28)]
        A, B = O.bicycle_AB(0.02, ini)
        A2, B2 = LQ.bicycle_AB(0.02, ini)
        assert np.allclose(A, A2, rtol=0, atol=1e-15) and np.array_equal(B, B2)
        M = int(rng.integers(0, 4))
        tgt = rng.uniform(-30, 30, 4); tw = rng.uniform(-2, 3, 4)
        aw = rng.uniform(0, 1, (2, M)); ot = rng.uniform(-30, 30, (M, 4)); ot[:, 3] = 0; ow = rng.uniform(0, 0.2, (M, 3))
        Q, q, R = O.cost_build(tgt, tw, 0.135, aw, ot, ow)
        Q2, q2, R2 = LQ.reach_avoid_cost(tgt, tw, 0.135, aw, ot, ow)
        assert np.array_equal(Q, Q2) and np.array_equal(q, q2) and np.array_equal(R, R2)


def test_quirks():
    """Q4: opponent-target diagonal assigns (erases -w_avoid); Q1: block-transposed LHS changes the answer
    (a textbook LHS would give a different u0), i.e. the quirk is observable and restated."""
    Q, q, R = O.cost_build([1, 2, 3, 4], [1, 1, 1, 1], 0.1, [[0.5], [0.5]], [[5, 6, 7, 0]], [[0.1, 0.2, 0.3]])
    assert Q[4, 4] == -0.1 and Q[5, 5] == -0.2 and Q[6, 6] == -0.3      # not -0.5-0.1
    assert Q[0, 4] == 0.5 and Q[0, 0] == -0.5 + 1.0
    assert q[7] == 0.0 and q[4] == 5 * -0.1
