#!/usr/bin/env python3
"""Emit tests/golden/step_mirror_fixtures.json: single-step golden vectors for SURVEY §8c (iii) produced by the INDEPENDENT
Python restatement oracle/step_numpy.py (written from the reference C#; it shares no code with the C oracle or the kernels).

The C oracle only supplies the STATES the vectors start from (a race has to be driven to somewhere): recorded hk_agent_state
records of a 4-agent Oval race at a few ticks.  Everything expected — players, heading-branch ids, initial / target states,
weights, u0, decoded controls, post-tick velocity / yaw rate / tire wear / pose — is computed by the mirror from those records.
tests/test_step_mirror.py then holds BOTH the C oracle (CPU test) and the HIP kernels (GPU test) to these values.
Run:  python tests/golden/make_step_mirror_fixtures.py"""
import base64
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np            # noqa: E402
import oracle_lib as O        # noqa: E402
from oracle import step_numpy as SN   # noqa: E402
from hierarchicalkarting_amd.config import make_config   # noqa: E402

SUITES = [
    # (make_config arguments, ticks at which states are recorded)
    (dict(num_envs=6, num_agents=4, jitter_seed=0x5EED0000), (76, 120, 300, 700, 1100, 1900, 2500, 3300)),                   # 2v2 Oval (configs[1])
    (dict(num_envs=6, num_agents=2, jitter_seed=0x5EED0000), (76, 200, 900, 2100)),                                           # 1v1 Oval: no 8 m filter, solve every tick
    (dict(num_envs=6, num_agents=4, jitter_seed=0x5EED0000, track="complex"), (76, 150, 400, 900, 1500, 2300, 3100, 4000)),   # every piece type of the track kit
]


def run_suite(cfg, ticks):
    b = make_config(**cfg)
    o = O.OracleEnv(b)
    o.reset()
    M = SN.Mirror(b)
    cadence = 4 if cfg["num_agents"] > 2 else 1
    cases, t = [], 0
    for snap in ticks:
        o.step(snap - t); t = snap
        while (t + 1) % cadence != 0:                    # the next tick must be a solve tick (HKA:317)
            o.step(1); t += 1
        before = o.agent_state().copy()
        es = o.env_state().copy()
        envs = []
        for env in range(o.E):
            games, after = M.solve_tick(before[env])
            envs.append({"games": games, "after": after})
        cases.append({"episode_step_before": int(es["episode_steps"][0]), "state_before_b64": base64.b64encode(before.tobytes()).decode(),
                      "env_state_before_b64": base64.b64encode(es.tobytes()).decode(), "envs": envs})
    return {"config": cfg, "record_bytes": int(before.dtype.itemsize), "cases": cases}


def main():
    suites = [run_suite(cfg, ticks) for cfg, ticks in SUITES]
    out = {"generator": "tests/golden/make_step_mirror_fixtures.py (oracle/step_numpy.py, the independent Python restatement)", "suites": suites}
    with open(os.path.join(HERE, "step_mirror_fixtures.json"), "w") as f:
        json.dump(out, f, indent=0)
    for s in suites:
        cases = s["cases"]
        ng = sum(g is not None for c in cases for e in c["envs"] for g in e["games"])
        na = sum(a is not None for c in cases for e in c["envs"] for a in e["after"])
        br = sorted({b for c in cases for e in c["envs"] for g in e["games"] if g for b in g["branch"]})
        npl = sorted({len(g["players"]) for c in cases for e in c["envs"] for g in e["games"] if g})
        print("%s: %d cases, %d games (player counts %s), %d free-motion karts, heading branches %s" % (s["config"], len(cases), ng, npl, na, br))


if __name__ == "__main__":
    main()
