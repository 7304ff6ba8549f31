"""csrc/hk_regroup_pos.h — where the regroup puts a lane group when it spreads the envs that hold multi-player games over the waves (round 6: the games
are solved in-wave, a pass at a time) — compiled for the host: the map is a bijection onto the class's slots for every (class size, hinted count) tried,
keeps each kind in rank order and spaces the hinted ones evenly.  The kernel that uses it (env_regroup_scatter_kernel) is covered on the GPU by every parity
test that crosses a regroup: a slot written twice or not at all loses an env."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spread_positions_are_a_bijection(tmp_path):
    exe = str(tmp_path / "regroup_pos_host_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "hierarchicalkarting_amd", "csrc"),
                           os.path.join(ROOT, "tests", "regroup_pos_host_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout[-500:] + out.stderr[-500:]
