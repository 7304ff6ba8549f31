"""Single-step golden vectors (SURVEY §8c iii; tests/golden/step_fixtures.json, emitted by make_step_fixtures.py):
restore a recorded full env state, run ONE solve tick, compare every ego's game (players, branch ids, targets, weights,
u0) and every kart's post-tick pose / velocity / yaw rate / tire wear / controls with the committed values — bit for bit.
The CPU test pins the oracle to the fixture, the GPU test pins the kernels (through the C ABI) to the same file."""
import base64
import json
import os
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.config import make_config

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_fixtures.json")
AGENT_DT = np.dtype(_lib.AgentState)
ENV_DT = np.dtype(_lib.EnvState)


def _replay(make_env):
    fx = json.load(open(FIX))
    assert fx["record_bytes"] == AGENT_DT.itemsize
    cfgkw = fx["config"]
    branches = set()
    for case in fx["cases"]:
        e = make_env(cfgkw)
        e.reset()
        st = np.frombuffer(base64.b64decode(case["state_before_b64"]), AGENT_DT).reshape(cfgkw["num_envs"], cfgkw["num_agents"]).copy()
        es = np.frombuffer(base64.b64decode(case["env_state_before_b64"]), ENV_DT).copy()
        e.set_agent_state(st)
        e.set_env_state(es)
        e.step(1)
        assert (e.env_state()["episode_steps"] % 4 == 0).all()          # it was a solve tick
        for gme in case["games"]:
            d = e.lq_debug(gme["env"], gme["ego"])
            n = gme["n_players"]
            assert d.n_players == n
            assert list(d.player_agent)[:n] == gme["player_agent"] and list(d.branch)[:n] == gme["branch"]
            for i in range(n):
                assert [float(v).hex() for v in d.initial[i]] == gme["initial"][i]
                assert [float(v).hex() for v in d.target[i]] == gme["target"][i]
                assert [float(v).hex() for v in d.target_w[i]] == gme["target_w"][i]
                assert float(d.control_w[i]).hex() == gme["control_w"][i]
            assert [float(v).hex() for v in d.u0] == gme["u0"]
            branches.update(gme["branch"])
        a = e.agent_state()
        for k, want in case["after"].items():
            if k in ("flags", "section_index"):
                assert a[k].tolist() == want, k
            else:
                got = [[float(v).hex() for v in row] for row in a[k].astype(np.float64)]
                assert got == want, k
    assert len(branches) >= 3


def test_oracle_reproduces_the_step_fixtures(monkeypatch):
    monkeypatch.setenv("HK_LQ_DEBUG", "1")
    _replay(lambda kw: O.OracleEnv(make_config(**kw)))


@pytest.mark.gpu
def test_kernels_reproduce_the_step_fixtures(monkeypatch):
    monkeypatch.setenv("HK_LQ_DEBUG", "1")
    import hierarchicalkarting_amd as hk
    _replay(lambda kw: hk.RacingEnv(hk.make_config(**kw)))
