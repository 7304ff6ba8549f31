"""tests/test_observation_moments.py on libhk, at a field twenty times the CPU test's (the same bands: tests/observation_moments.py)."""
import pytest
import hierarchicalkarting_amd as hk
import observation_moments as M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", sorted(M.ACTORS))
def test_libhk_observation_moments_match_the_actors_normaliser(model):
    stack, A, _ = M.ACTORS[model]
    m, s, n = M.moments(hk.RacingEnv, model, E=512 if A == 2 else 256, ticks=2400)
    rows = M.compare(model, m, s)
    assert not [r for r in rows if r[5] == "OUT"], M.report(model, rows, n)
    rays = [abs(r[3]) for r in rows if r[0] == "rays"]
    assert max(rays) < 0.5 and sum(rays) / len(rays) < 0.3, M.report(model, rows, n)
