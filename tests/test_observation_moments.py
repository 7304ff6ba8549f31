"""The observation stream's per-dimension moments against the input normalisers of the reference's four trained actors (tests/observation_moments.py);
CPU oracle.  tests/test_observation_moments_gpu.py is the same comparison on libhk at a larger field."""
import pytest
import oracle_lib as O
import observation_moments as M


@pytest.mark.parametrize("model", sorted(M.ACTORS))
def test_oracle_observation_moments_match_the_actors_normaliser(model):
    stack, A, _ = M.ACTORS[model]
    m, s, n = M.moments(O.OracleEnv, model, E=24 if A == 2 else 12, ticks=1600)
    rows = M.compare(model, m, s)
    out = [r for r in rows if r[5] == "OUT"]
    assert not out, M.report(model, rows, n)
    # the rays are the block the scene's Sensors[] order was recovered from: every one of them within a third of a standard deviation of the mean ... almost
    rays = [abs(r[3]) for r in rows if r[0] == "rays"]
    assert max(rays) < 0.5 and sum(rays) / len(rays) < 0.3, M.report(model, rows, n)
    # every residual is present (a residual that vanished means the table is stale)
    assert {(r[0], r[1]) for r in rows if r[5].startswith("residual")} == set(M.RESIDUALS) if A > 1 else True
