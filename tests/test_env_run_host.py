"""The tick kernel's device code on the host, under AddressSanitizer + UndefinedBehaviorSanitizer (tests/env_run_host_check.cpp through the
stand-in <hip/hip_runtime.h> in tests/host_emu: a lane group is four threads, every cross-lane primitive a barrier-guarded exchange that
fails when only part of the group reaches it).  The GPU pool has no sanitizer (GPU ASan is refused there), and round 2's Training-mode
failure that moved with the loop's source form was never explained — this is the check that can see an out-of-bounds table read, an
uninitialised flag or a group exchange inside divergent control flow in that code, whatever the optimiser makes of it.

The run must also be RIGHT: every field of every agent record equals the C oracle's at each chunk end (multi-player games are the solver
kernel's part; the driver plays it with the oracle's recorded controls, see the .cpp header)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib as HL
from hierarchicalkarting_amd.config import make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hierarchicalkarting_amd", "csrc")


def _build(tmp_path_factory, name, *defs):
    exe = str(tmp_path_factory.mktemp("emu") / name)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-ffp-contract=off", "-Wno-attributes", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", *defs, "-I" + os.path.join(ROOT, "tests", "host_emu"), "-I" + CSRC,
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "env_run_host_check.cpp"), "-o", exe, "-lpthread"])
    return exe


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    return _build(tmp_path_factory, "env_run_host_check")


@pytest.fixture(scope="module")
def harness_ifelse(tmp_path_factory):
    """the loop end written as a chain of branches (hk_env_run.h, HK_LOOP_IFELSE): the source form round 2's Training-mode build failed with"""
    return _build(tmp_path_factory, "env_run_host_check_ifelse", "-DHK_LOOP_IFELSE=1")


@pytest.fixture(scope="module")
def harness_g8(tmp_path_factory):
    """the 8-lane groups (hk::g8, configs[4]): eight host threads per race instance"""
    return _build(tmp_path_factory, "env_run_host_check_g8", "-DHK_GA=8", "-DHK_GA_NS=g8", "-DHK_EMU_LANES=8")


def _run(harness, tmp_path, built, n_ticks, chunk, eager, run_cap):
    E, A = built.cfg.num_envs, built.cfg.num_agents
    o = O.OracleEnv(built)
    o.reset()
    ctl = np.zeros((n_ticks, E, A), np.dtype([("flags", "<u4"), ("steering", "<f4")]))
    want = []
    for t in range(n_ticks):
        o.step(1)
        st = o.agent_state()
        ctl[t]["flags"], ctl[t]["steering"] = st["flags"], st["steering"]
        if (t + 1) % chunk == 0 or t + 1 == n_ticks:
            want.append((t + 1, st, o.env_state()))
    L, NW = built.cfg.num_sections, built.cfg.num_walls
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([0x484b4531, C.sizeof(HL.Config), L, NW, n_ticks, chunk, eager, E, A, run_cap], "<i4").tobytes())
        f.write(bytes(built.cfg))
        f.write(bytes(built.sections)[:L * C.sizeof(HL.Section)])
        f.write(bytes(built.walls)[:NW * C.sizeof(HL.WallSeg)])
        f.write(ctl.tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([harness, fin, fout], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    raw = np.fromfile(fout, np.uint8)
    per = E * A * O.AGENT_DT.itemsize + E * O.ENV_DT.itemsize
    assert raw.size == per * len(want)
    for k, (t, ast, est) in enumerate(want):
        blk = raw[k * per:(k + 1) * per]
        ga = blk[:E * A * O.AGENT_DT.itemsize].view(O.AGENT_DT).reshape(E, A)
        ge = blk[E * A * O.AGENT_DT.itemsize:].view(O.ENV_DT)
        for name in ga.dtype.names:
            if not np.array_equal(ga[name], ast[name]):
                bad = np.argwhere(ga[name] != ast[name])
                e, a = bad[0][0], bad[0][1]
                raise AssertionError("tick %d field %s env %d agent %d: emulation %r oracle %r (%d mismatches)" % (
                    t, name, e, a, ga[name][e, a], ast[name][e, a], len(bad)))
        for name in ("episode_steps", "inactive_mask", "experiment_num", "episodes_done", "status", "initial_started"):
            assert np.array_equal(ge[name], est[name]), (t, name, ge[name], est[name])
    stats = dict(zip(r.stdout.split()[0::2], map(int, r.stdout.split()[1::2])))
    assert stats["status"] == 0
    return stats


@pytest.mark.parametrize("eager,run_cap,chunk", [(0, 8, 1), (1, 32, 20), (0, 4, 7)])
def test_tick_kernel_device_code_is_clean_under_sanitizers(harness, tmp_path, eager, run_cap, chunk):
    """4 agents, 3 envs with start-grid jitter, an episode that times out and restarts inside the run (max_episode_steps 260), launches of
    one tick / eager assembly with the long-call tick budget / short budget with ragged chunks."""
    built = make_config(3, 4, jitter_seed=11, max_episode_steps=260)
    stats = _run(harness, tmp_path, built, 330, chunk, eager, run_cap)
    assert stats["queued_games"] > 100          # the start grid: packs of 2-4 players


def test_complex_track_long_run_under_sanitizers(harness, tmp_path):
    """the Complex track (41 sections, 842 wall segments: curves of 32 one-metre segments), 6 envs, two episodes with a restart in between"""
    built = make_config(6, 4, track="complex", jitter_seed=23, max_episode_steps=420)
    stats = _run(harness, tmp_path, built, 600, 40, 1, 12)
    assert stats["queued_games"] > 200


def test_rewarded_handle_under_sanitizers(harness, tmp_path):
    built = make_config(2, 4, jitter_seed=5, max_episode_steps=220, rewards=1)
    _run(harness, tmp_path, built, 260, 10, 1, 32)


def test_three_agents_under_sanitizers(harness, tmp_path):
    """3 agents in a quad: the fourth lane is idle and takes part in every group exchange"""
    built = make_config(2, 3, jitter_seed=3, max_episode_steps=200)
    _run(harness, tmp_path, built, 230, 9, 1, 32)


def _training(**kw):
    return make_config(2, 4, env_mode=HL.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=150, rewards=1,
                       jitter_seed=0, **kw)


def test_training_mode_instantiation_under_sanitizers(harness, tmp_path):
    """env_run_kernel<true, true, true>: random scatter resets inside the tick loop, planRandomly, rewards — the instantiation whose parity
    depended on the loop's source form in round 2"""
    _run(harness, tmp_path, _training(), 330, 11, 1, 32)


def test_training_mode_with_the_other_loop_form(harness_ifelse, tmp_path):
    _run(harness_ifelse, tmp_path, _training(track="complex"), 200, 50, 1, 32)


def test_eight_lane_groups_under_sanitizers(harness_g8, tmp_path):
    """hk::g8 (configs[4]: 8 agents, two teams of four): the start grid queues games of up to 8 players; an episode restarts inside the run"""
    built = make_config(2, 8, jitter_seed=7, max_episode_steps=240)
    stats = _run(harness_g8, tmp_path, built, 300, 25, 0, 8)
    assert stats["queued_games"] > 100


def test_six_agents_in_eight_lane_groups_under_sanitizers(harness_g8, tmp_path):
    """6 agents in a group of 8: two idle lanes take part in every group exchange"""
    built = make_config(2, 6, jitter_seed=9, max_episode_steps=200)
    _run(harness_g8, tmp_path, built, 230, 10, 0, 8)
