"""The reference's experiment set-ups through libhk.

All 22 set-ups of tests/golden/reference_experiments.json — the reference's own trained actors on the f32 MFMA, the MCTS planner,
the LQNG solver, both tracks, 1v1 and 2v2 — run on the GPU for their full 48 / 50 races; every race's hk_episode_result must equal
the CPU oracle's bit for bit (the hashes of tests/golden/experiment_oracle.json, which tests/test_reference_logs.py holds the
oracle to and bands against the reference's ExperimentLogs).  One statistics pass therefore serves both sides."""
import json
import os
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
ORA = json.load(open(os.path.join(ROOT, "tests", "golden", "experiment_oracle.json")))


@pytest.mark.parametrize("name", sorted(ORA))
def test_libhk_races_equal_the_oracles(name):
    import compare_experiment_logs as CE
    import hierarchicalkarting_amd as hk
    res, stats = CE.run_ours(name, hk.RacingEnv, ORA[name]["mcts_iterations"])
    assert stats == ORA[name]["stats"], name
    assert CE.results_hash(res) == ORA[name]["results_sha256"], name
