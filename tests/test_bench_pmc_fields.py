"""bench.py's roofline extras come from the committed PMC summary (profiles/r06_pmc_summary.json): the summary must describe THIS tree's kernel sources
(otherwise the bench line carries `traffic: null`), and the derived figures must be what their definitions say.  No GPU, no libhk."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("hk_bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_summary_matches_the_kernel_sources():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary.json")))
    assert d["sources_sha16"] == source_hash(), "the kernels changed after the PMC passes: rerun tools/experiments/evidence.sh and commit its summary"


def test_traffic_and_binding_fields():
    b = _bench()
    traffic, prov, binding = b.pmc_fields("env_run_kernel", 131072.0)
    assert traffic and 2.0e7 < traffic < 1.0e8 and not prov.get("stale")
    assert 0.2 < binding["wave_issuing_valu_frac"] < 0.5 and 40 < binding["valu_lanes_active_of_64"] <= 64
    # a launch alone on the GPU (as the PMC passes run it) is 2 048 waves on 1 024 SIMDs
    assert abs(binding["simd_valu_busy_frac_launch_alone"] - 2.0 * binding["wave_issuing_valu_frac"]) < 1e-12
    cad = b.cadence_traffic(131072.0)
    parts = cad["parts_bytes_per_launch"]
    assert abs(cad["bytes_per_cadence_launch_set"] - sum(parts.values())) < 1.0
    assert 0.3 < cad["ratio_to_algorithmic"] < 1.5


def test_valu_port_use_is_instructions_over_wall():
    b = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary.json")))
    # the steady state's launch set: tick + B1 (in-wave solves), + the spread solver's launch by its share of rounds
    w = min(1.0, d["lqn_spread_kernel"]["launches_seen"] / d["env_run_kernel"]["launches_seen"]) if "lqn_spread_kernel" in d else 0.0
    quad = d["env_run_kernel"]["sq"]["SQ_ACTIVE_INST_VALU"] + d["env_b1_kernel"]["sq"]["SQ_ACTIVE_INST_VALU"] + (d["lqn_spread_kernel"]["sq"]["SQ_ACTIVE_INST_VALU"] * w if w else 0.0)
    v = b.valu_port_use(1538, 0.090)
    assert abs(v["frac"] - quad * 4.0 * 1538 / (1024 * 2.4e9 * 0.090)) < 1e-9
    assert 0.4 < v["frac"] < 0.95           # (the round's protocol window: 0.76)
    assert b.valu_port_use(1538, 0.0) is None
