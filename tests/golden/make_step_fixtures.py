#!/usr/bin/env python3
"""Emit tests/golden/step_fixtures.json: single-step golden vectors for SURVEY §8c (iii), produced by the CPU oracle.

  a4 (HierarchicalKartAgent.SolveLQR): for recorded full env states, what every ego's game looked like on the next solve
      tick — players, heading-heuristic branch ids, initial / target states, weights, u0 — and the decoded controls.
  a6 (ArcadeKart.MoveVehicle + engine step): state before one tick -> pose, velocity, yaw rate, tire wear after it.
Floats are stored as exact hex strings (float.hex), the env state as the raw hk_agent_state bytes (base64).
Run:  python tests/golden/make_step_fixtures.py        (rewrites the fixture; the tests then pin the oracle AND the kernels to it)"""
import base64
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np            # noqa: E402
import oracle_lib as O        # noqa: E402
from hierarchicalkarting_amd.config import make_config   # noqa: E402

CONFIG = dict(num_envs=3, num_agents=4, jitter_seed=0x5EED0000)
SNAP_TICKS = (76, 300, 1100, 2500)      # start of the race (everyone within 8 m), spread field, later laps


def fh(x):
    return float(x).hex()


def main():
    os.environ["HK_LQ_DEBUG"] = "1"
    b = make_config(**CONFIG)
    o = O.OracleEnv(b)
    o.reset()
    cases = []
    t = 0
    for snap in SNAP_TICKS:
        o.step(snap - t); t = snap                       # episode step `snap` done; the next tick (snap + 1) ...
        while (t + 1) % 4 != 0:                          # ... must be a solve tick: episode_steps % 4 == 0 (HKA:317)
            o.step(1); t += 1
        before = o.agent_state().copy()
        es = o.env_state().copy()
        o.step(1); t += 1
        after = o.agent_state()
        games = []
        for env in range(o.E):
            for ego in range(o.A):
                d = o.lq_debug(env, ego)
                n = d.n_players
                games.append({"env": env, "ego": ego, "n_players": n, "player_agent": list(d.player_agent)[:n], "branch": list(d.branch)[:n],
                              "initial": [[fh(v) for v in d.initial[i]] for i in range(n)],
                              "target": [[fh(v) for v in d.target[i]] for i in range(n)],
                              "target_w": [[fh(v) for v in d.target_w[i]] for i in range(n)],
                              "control_w": [fh(d.control_w[i]) for i in range(n)], "u0": [fh(v) for v in d.u0]})
        moved = {k: [[fh(v) for v in row] for row in after[k].astype(np.float64)] for k in ("px", "pz", "yaw", "vx", "vz", "wy", "acc_ang_v", "steering")}
        moved["flags"] = after["flags"].tolist()
        moved["section_index"] = after["section_index"].tolist()
        cases.append({"episode_step_before": int(es["episode_steps"][0]), "state_before_b64": base64.b64encode(before.tobytes()).decode(),
                      "env_state_before_b64": base64.b64encode(es.tobytes()).decode(), "games": games, "after": moved})
    out = {"generator": "tests/golden/make_step_fixtures.py (CPU oracle)", "config": CONFIG, "record_bytes": int(before.dtype.itemsize),
           "cases": cases}
    with open(os.path.join(HERE, "step_fixtures.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %d cases, %d games each" % (len(cases), len(cases[0]["games"])))


if __name__ == "__main__":
    main()
