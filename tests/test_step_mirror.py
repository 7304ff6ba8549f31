"""Single-step vectors from the INDEPENDENT Python restatement (oracle/step_numpy.py, written from the reference C#):
tests/golden/step_mirror_fixtures.json holds recorded kart states and, computed by the mirror alone, what one solve tick makes
of them — every ego's game (players, heading-branch ids, initial / target states, weights, u0), the decoded controls and the
post-tick velocity / yaw rate / tire wear / pose of every kart in free motion.  The CPU test holds the C oracle to these
values, the GPU test the HIP kernels (through the C ABI).

Tolerances (the mirror computes in float64 with numpy transcendentals; the product in fp32 physics / fp64 Riccati with the pinned
detmath): discrete outcomes exact; states, targets, weights 2e-6 relative; u0 1e-4 relative (the Riccati recursion amplifies
input rounding) — the north-star bound is 1e-4 fp32; post-tick kart fields 2e-5."""
import base64
import json
import os
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.config import make_config

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_mirror_fixtures.json")
AGENT_DT = np.dtype(_lib.AgentState)
ENV_DT = np.dtype(_lib.EnvState)


def _close(got, want, rel, abs_=1e-6):
    got, want = np.asarray(got, float), np.asarray(want, float)
    return np.all(np.abs(got - want) <= abs_ + rel * np.abs(want))


def _replay(make_env):
    branches, n_games, n_free, n_players = set(), 0, 0, set()
    for suite in json.load(open(FIX))["suites"]:
        assert suite["record_bytes"] == AGENT_DT.itemsize
        kw = suite["config"]
        cadence = 4 if kw["num_agents"] > 2 else 1
        for case in suite["cases"]:
            e = make_env(kw)
            e.reset()
            st = np.frombuffer(base64.b64decode(case["state_before_b64"]), AGENT_DT).reshape(kw["num_envs"], kw["num_agents"]).copy()
            es = np.frombuffer(base64.b64decode(case["env_state_before_b64"]), ENV_DT).copy()
            e.set_agent_state(st)
            e.set_env_state(es)
            e.step(1)
            assert (e.env_state()["episode_steps"] % cadence == 0).all()
            a = e.agent_state()
            for env, rec in enumerate(case["envs"]):
                for ego, g in enumerate(rec["games"]):
                    if g is None:
                        continue
                    d = e.lq_debug(env, ego)
                    n = len(g["players"])
                    where = (kw.get("track", "oval"), kw["num_agents"], case["episode_step_before"], env, ego)
                    assert d.n_players == n and list(d.player_agent)[:n] == g["players"], where
                    assert list(d.branch)[:n] == g["branch"], (where, list(d.branch)[:n], g["branch"])
                    for i in range(n):
                        assert _close(list(d.initial[i]), g["initial"][i], 2e-6), (where, i, "initial")
                        assert _close(list(d.target[i])[:3], g["target"][i][:3], 2e-6), (where, i, "target")
                        assert _close(d.target[i][3], g["target"][i][3], 2e-6, 2e-6), (where, i, "target heading", d.target[i][3], g["target"][i][3])
                        assert _close(list(d.target_w[i]), g["target_w"][i], 2e-6, 1e-12), (where, i, "weights")
                        assert _close(d.control_w[i], g["control_w"][i], 1e-12), (where, i)
                    assert _close(list(d.u0), g["u0"], 1e-4, 1e-5), (where, list(d.u0), g["u0"])
                    fl = int(a["flags"][env, ego])
                    assert bool(fl & _lib.HK_F_ACCEL) == g["accelerate"] and bool(fl & _lib.HK_F_BRAKE) == g["brake"], where
                    assert _close(a["steering"][env, ego], g["steering"], 1e-4, 1e-5), (where, "steering")
                    branches.update(g["branch"]); n_games += 1; n_players.add(n)
                for k, m in enumerate(rec["after"]):
                    if m is None:
                        continue
                    for name, want in m.items():
                        got = float(a[name][env, k])
                        where = (kw.get("track", "oval"), kw["num_agents"], case["episode_step_before"], env, k, name, got, want)
                        if name == "yaw":
                            dy = abs(got - want); dy = min(dy, abs(dy - 2 * np.pi))
                            assert dy <= 2e-5, where
                        else:
                            assert _close(got, want, 2e-5, 2e-5), where
                    n_free += 1
    assert len(branches) >= 5 and n_games >= 350 and n_free >= 300 and n_players >= {1, 2}, (branches, n_games, n_free, n_players)


def test_oracle_matches_the_independent_mirror(monkeypatch):
    monkeypatch.setenv("HK_LQ_DEBUG", "1")
    _replay(lambda kw: O.OracleEnv(make_config(**kw)))


@pytest.mark.gpu
def test_kernels_match_the_independent_mirror(monkeypatch):
    monkeypatch.setenv("HK_LQ_DEBUG", "1")
    import hierarchicalkarting_amd as hk
    _replay(lambda kw: hk.RacingEnv(hk.make_config(**kw)))


def test_mirror_regenerates_its_own_fixture():
    """the committed file IS what oracle/step_numpy.py computes today from the recorded states (no hand edits, no drift)"""
    from oracle import step_numpy as SN
    for suite in json.load(open(FIX))["suites"]:
        kw = suite["config"]
        M = SN.Mirror(make_config(**kw))
        for case in suite["cases"][::3]:
            st = np.frombuffer(base64.b64decode(case["state_before_b64"]), AGENT_DT).reshape(kw["num_envs"], kw["num_agents"])
            for env in (0, kw["num_envs"] - 1):
                games, after = M.solve_tick(st[env])
                assert json.loads(json.dumps({"games": games, "after": after})) == case["envs"][env]
