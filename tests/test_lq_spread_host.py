"""The lane-per-(player, row) solver of 2-, 3- and 4-player games (csrc/hk_lq_spread.h) compiled for the host — one thread per lane of a
game's lane set, four 2-player games side by side as in a wave, the wave-level LDS ordering a barrier — against the C oracle's
solveFeedbackLQR on random games with the structure SolveLQR produces (compact reach-avoid cost rows, linearised bicycles): the controls
must be identical bit for bit.  (The same header runs on the GPU — the solver launch of a spread field and the tail of the B1 kernel — where
the env parity tests compare whole trajectories; this test localises an arithmetic slip without a GPU.)"""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
from oracle import lq_numpy as LQ

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GP_NO = 3
GP_X0, GP_A4, GP_TW, GP_TGT, GP_RC, GP_AW = 0, 4, 8, 12, 16, 17
GP_OPW = GP_AW + GP_NO; GP_OPT = GP_OPW + 3 * GP_NO; GP_M = GP_OPT + 3 * GP_NO; GP_FIELDS = (GP_M + 2) & ~1


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lqs") / "lq_spread_host_check")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "hierarchicalkarting_amd", "csrc"),
                           os.path.join(ROOT, "tests", "lq_spread_host_check.cpp"), "-o", exe, "-lpthread"])
    return exe


def _games(rng, N, ng, inactive_others=False):
    """-> (GameSoA doubles [N * GP_FIELDS][ng], the oracle's controls): per player the compact fields assemble_player writes and the dense
    cost of KartLQRCosts.cs:57-127 in the player's own order [k, the others]"""
    d = np.zeros((N * GP_FIELDS, ng))
    dt = float(np.float32(0.02))
    near = max(N - 1, 1)
    ref = []
    for g in range(ng):
        As, Bs, Qs, qs, Rs, x0 = [], [], [], [], [], []
        for i in range(N):
            x, z, v, th = rng.uniform(-50, 25), rng.uniform(-50, 65), rng.uniform(0, 15), rng.uniform(0, 2 * np.pi)
            ini = [float(np.float32(t)) for t in (x, z, v, th)]
            a4 = [np.cos(ini[3]) * dt, np.sin(ini[3]) * dt, -np.sin(ini[3]) * dt * ini[2], np.cos(ini[3]) * dt * ini[2]]
            A = np.eye(4); A[0, 2], A[1, 2], A[0, 3], A[1, 3] = a4
            B = np.zeros((4, 2)); B[2, 0] = dt; B[3, 1] = dt
            slow = v <= 5
            hw = (2.5 * near) if N > 2 else 1.9
            tw = [near * 0.93, near * 0.93, near * -2.0, hw] if slow else [near * 0.93 / max(1, v), near * 0.93 / max(1, v), near * 5e-4, hw]
            tgt = [x + rng.uniform(-10, 10), z + rng.uniform(-10, 10), 0.0 if slow else 15.0, th + rng.uniform(-.6, .6)]
            M = N - 1
            rc = 0.135 if N > 2 else 0.115
            f = lambda k: i * GP_FIELDS + k
            for c in range(4):
                d[f(GP_X0 + c), g] = ini[c]; d[f(GP_A4 + c), g] = a4[c]; d[f(GP_TW + c), g] = tw[c]; d[f(GP_TGT + c), g] = tgt[c]
            d[f(GP_RC), g] = rc; d[f(GP_M), g] = M
            aw = np.zeros((2, M)); ot = np.zeros((M, 4)); ow = np.zeros((M, 3))
            for j in range(M):
                far = inactive_others and rng.random() < 0.6
                w = 0.0 if far else 1.0 / (rng.uniform(1, 8) ** 1.5 * rng.choice([0.55, 1.7, 0.45, 1.3]) / near)
                opw = [0.0] * 3 if far else [0.1 / (max(1, v) * near)] * 2 + [0.08 / near]
                opt = [rng.uniform(-50, 25), rng.uniform(-50, 65), 15.0]
                d[f(GP_AW + j), g] = w
                for c in range(3):
                    d[f(GP_OPW + 3 * j + c), g] = opw[c]; d[f(GP_OPT + 3 * j + c), g] = opt[c]
                aw[:, j] = w; ot[j, :3] = opt; ow[j] = opw
            Q, q, R = LQ.reach_avoid_cost(tgt, tw, rc, aw, ot, ow)
            As.append(A); Bs.append(B); Qs.append(Q); qs.append(q); Rs.append(R); x0 += ini
        ref.append(O.lq_solve(np.array(As), np.array(Bs), np.array(Qs), np.array(qs), np.array(Rs), np.array(x0), 3))
    return d, ref


@pytest.mark.parametrize("N,serial_w", [(2, 0), (3, 0), (4, 0), (4, 1), (2, 1)])
@pytest.mark.parametrize("inactive_others", [False, True])
def test_spread_solver_arithmetic_equals_the_oracle(harness, N, serial_w, inactive_others):
    """serial_w: the players take one W block in turn (the form env_b1_kernel's waves use for 4-player games, 5 KB of LDS instead of 11)"""
    rng = np.random.default_rng(500 + 10 * N + inactive_others)
    ng = 10 if N == 2 else 3           # (N = 2: two full groups of four games and a ragged last one)
    d, ref = _games(rng, N, ng, inactive_others)
    inp = "%d %d %d\n" % (N, ng, serial_w) + "\n".join(repr(float(x)) for x in d.ravel())
    out = subprocess.run([harness], input=inp, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == ng
    for g, line in enumerate(out):
        a, b, sing = line.split()
        assert float.fromhex(a) == ref[g][0] and float.fromhex(b) == ref[g][1] and int(sing) == 0, (g, a, ref[g][0], b, ref[g][1])
