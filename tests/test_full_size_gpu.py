"""BASELINE.json's full size (configs[1]: 4-agent Oval, 65 536 envs on one GPU), checked through size-independent properties:
an env's trajectory depends only on its global id (windows of the big batch equal small handles started at those ids, and —
for a few of them — the CPU oracle), two runs of the big batch give the same checksum of checksums, and the records stay sane."""
import hashlib
import numpy as np
import pytest
import oracle_lib as O

pytestmark = pytest.mark.gpu

E_FULL = 65536
SEED = 0x5EED0000
CHUNKS = (80, 200, 320)          # 600 ticks: start hold, the 4-player games of the race start, the field spreading out


def _digest(st):
    """checksum of per-env checksums (order-sensitive)"""
    raw = np.ascontiguousarray(st).view(np.uint8).reshape(st.shape[0], -1)
    per_env = [hashlib.sha256(raw[e].tobytes()).digest() for e in range(0, st.shape[0], 257)]      # every 257th env: 256 digests
    return hashlib.sha256(b"".join(per_env) + hashlib.sha256(raw.tobytes()).digest()).hexdigest()


def test_full_batch_windows_match_small_handles_and_the_oracle():
    import hierarchicalkarting_amd as hk
    big = hk.RacingEnv(hk.make_config(E_FULL, 4, jitter_seed=SEED))
    big.reset()
    windows = (0, 30000, E_FULL - 64)
    small = [hk.RacingEnv(hk.make_config(64, 4, jitter_seed=SEED, env_id_base=b)) for b in windows]
    for s in small:
        s.reset()
    ob = hk.make_config(8, 4, jitter_seed=SEED, env_id_base=30000)
    orc = O.OracleEnv(ob)
    orc.reset()
    digests = []
    prev_sec = None
    for n in CHUNKS:
        big.step(n)
        st = big.agent_state()
        digests.append(_digest(st))
        for b, s in zip(windows, small):
            s.step(n)
            ss = s.agent_state()
            for name in st.dtype.names:
                assert np.array_equal(st[name][b:b + 64], ss[name]), (n, b, name)
        orc.step(n)
        os_ = orc.agent_state()
        for name in st.dtype.names:
            x, y = st[name][30000:30008], os_[name]
            assert np.array_equal(x, y), (n, name)
        # sanity over the whole batch
        for name in ("px", "pz", "vx", "vz", "yaw", "wy"):
            assert np.isfinite(st[name]).all(), name
        assert (np.abs(st["px"]) < 200).all() and (np.abs(st["pz"]) < 200).all()
        if prev_sec is not None:
            assert (st["section_index"] >= prev_sec).all()          # nobody drives backwards through a checkpoint here
        prev_sec = st["section_index"].copy()
    assert prev_sec.min() >= 1 and prev_sec.max() <= 24      # ~150 m into the first lap of a 24-section track
    # the same batch again: bit-identical (checksum of checksums)
    again = hk.RacingEnv(hk.make_config(E_FULL, 4, jitter_seed=SEED))
    again.reset()
    for n, d in zip(CHUNKS, digests):
        again.step(n)
        assert _digest(again.agent_state()) == d


@pytest.mark.parametrize("N", [2, 4])
def test_lq_batch_at_full_size_is_a_pure_per_game_function(N):
    """hk_lq_solve_batch with 262 144 games (65 536 envs x 4 egos): a game's solution does not depend on the batch it sits in
    nor on its place in it — a permuted tiling of 512 random games returns the (oracle-checked) solutions of those 512, bit for bit."""
    import hierarchicalkarting_amd as hk
    from oracle import lq_numpy as LQ
    rng = np.random.default_rng(1234 + N)
    base = [LQ.random_game(rng, N) for _ in range(512)]
    args = [np.array([g[k] for g in base]) for k in range(6)]
    u_small = hk.solve_feedback_lqr_batch(*args, 3)
    for g in range(0, 512, 37):                                      # anchor a few on the C oracle
        uo = O.lq_solve(*[a[g] for a in args], 3)
        assert np.array_equal(np.asarray(uo).view(np.uint64), u_small[g].view(np.uint64)), g
    B = E_FULL * 4
    idx = np.random.default_rng(7).integers(0, 512, B)
    u_big = hk.solve_feedback_lqr_batch(*[a[idx] for a in args], 3)
    assert u_big.shape == (B, 2)
    assert np.array_equal(u_big.view(np.uint64), u_small[idx].view(np.uint64))


def test_planner_batch_of_configs2_size_windows_match_small_handles_and_the_oracle():
    """BASELINE.json configs[2] at its full size — 4-agent Complex track, every agent MCTS-LQNG, 16 384 envs — through the same
    size-independent property: windows of the big batch equal small handles started at those env ids, and the CPU oracle, in planner
    state and kart records.  The big handle steps in long calls (the planner's stretch-wise schedule with envs pausing for their
    searches), the small ones tick by tick in uneven chunks (the deadline schedule): the schedules must not show."""
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd import _lib
    MC = _lib.HK_HIGH_MCTS
    kw = dict(track="complex", high_mode=[MC] * 4, tree_search_depth=8, mcts_iterations=24, jitter_seed=SEED)
    E = 16384
    big = hk.RacingEnv(hk.make_config(E, 4, **kw))
    big.reset()
    windows = (0, 9000, E - 32)
    small = [hk.RacingEnv(hk.make_config(32, 4, env_id_base=b, **kw)) for b in windows]
    for s in small:
        s.reset()
    orc = O.OracleEnv(hk.make_config(6, 4, env_id_base=9000, **kw))
    orc.reset()
    for n in (120, 130):                                    # across the replans at ticks 100 and 200 (root reuse at 100)
        big.step(n)
        for s in small:
            for c in [30] * (n // 30) + ([n % 30] if n % 30 else []):        # <= 32 ticks per call: the deadline schedule
                s.step(c)
        orc.step(n)
        st, ms = big.agent_state(), big.mcts_state()
        for b, s in zip(windows, small):
            ss, sm = s.agent_state(), s.mcts_state()
            for name in st.dtype.names:
                assert np.array_equal(st[name][b:b + 32], ss[name]), (n, b, name)
            for name in ("searches", "ready_step", "root_live", "root_cycles", "root_phases", "sec_time"):
                assert np.array_equal(ms[name][b:b + 32], sm[name]), (n, b, name)
            for name in ms["best"].dtype.names:
                assert np.array_equal(ms["best"][name][b:b + 32], sm["best"][name]), (n, b, name)
        os_, om = orc.agent_state(), orc.mcts_state()
        for name in st.dtype.names:
            assert np.array_equal(st[name][9000:9006], os_[name]), (n, name)
        for name in ms["best"].dtype.names:
            assert np.array_equal(ms["best"][name][9000:9006], om["best"][name]), (n, name)
    assert (ms["searches"] == 3).all() and (ms["root_phases"][:, :] >= 1).all()
