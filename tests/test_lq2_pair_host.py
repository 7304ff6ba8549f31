"""The two-lane solver of 2-player games (csrc/hk_lq2_pair.h) compiled for the host — two threads play the lanes of a pair, the DPP
exchange is a barrier-guarded swap — against the C oracle's solveFeedbackLQR on random games with the structure SolveLQR produces:
the controls must be identical bit for bit.  (The same header runs on the GPU, where tests/test_env_gpu.py & co. compare whole
trajectories; this test localises an arithmetic slip without a GPU.)"""
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GP_NO = 3
GP_X0, GP_A4, GP_TW, GP_TGT, GP_RC, GP_AW = 0, 4, 8, 12, 16, 17
GP_OPW = GP_AW + GP_NO; GP_OPT = GP_OPW + 3 * GP_NO; GP_M = GP_OPT + 3 * GP_NO; GP_FIELDS = (GP_M + 2) & ~1


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lq2") / "lq2_pair_host_check")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-I" + os.path.join(ROOT, "hierarchicalkarting_amd", "csrc"),
                           os.path.join(ROOT, "tests", "lq2_pair_host_check.cpp"), "-o", exe, "-lpthread"])
    return exe


def _games(rng, ng, inactive_other=False):
    d = np.zeros((2 * GP_FIELDS, ng))
    dt = float(np.float32(0.02))
    ref = []
    for g in range(ng):
        As, Bs, Qs, qs, Rs, x0 = [], [], [], [], [], []
        for i in range(2):
            x, z, v, th = rng.uniform(-50, 25), rng.uniform(-50, 65), rng.uniform(0, 15), rng.uniform(0, 2 * np.pi)
            ini = [float(np.float32(t)) for t in (x, z, v, th)]
            a4 = [np.cos(ini[3]) * dt, np.sin(ini[3]) * dt, -np.sin(ini[3]) * dt * ini[2], np.cos(ini[3]) * dt * ini[2]]
            A = np.eye(4); A[0, 2], A[1, 2], A[0, 3], A[1, 3] = a4
            B = np.zeros((4, 2)); B[2, 0] = dt; B[3, 1] = dt
            slow = v <= 5
            tw = [0.93, 0.93, -2.0, 1.9] if slow else [0.93 / max(1, v), 0.93 / max(1, v), 5e-4, 1.9]
            tgt = [x + rng.uniform(-10, 10), z + rng.uniform(-10, 10), 0.0 if slow else 15.0, th + rng.uniform(-.6, .6)]
            aw0 = 0.0 if inactive_other else 1.0 / (rng.uniform(1, 8) ** 1.5 * (0.45 if i == 0 else 1.3))
            opw = [0.0] * 3 if inactive_other else [0.1 / max(1, v)] * 2 + [0.08]
            opt = [rng.uniform(-50, 25), rng.uniform(-50, 65), 15.0]
            rc = 0.115
            f = lambda k: i * GP_FIELDS + k
            for c in range(4):
                d[f(GP_X0 + c), g] = ini[c]; d[f(GP_A4 + c), g] = a4[c]; d[f(GP_TW + c), g] = tw[c]; d[f(GP_TGT + c), g] = tgt[c]
            d[f(GP_RC), g] = rc; d[f(GP_AW), g] = aw0; d[f(GP_M), g] = 1
            for c in range(3):
                d[f(GP_OPW + c), g] = opw[c]; d[f(GP_OPT + c), g] = opt[c]
            # the dense cost of KartLQRCosts.cs:57-127 in the player's own order [k, the other]
            Q = np.zeros((8, 8)); q = np.zeros(8)
            total = 0.0 - aw0
            for s in range(4):
                Q[s, s] = (total if s < 2 else 0.0) + tw[s]
                if s < 2:
                    Q[s, 4 + s] = aw0; Q[4 + s, s] = aw0
                q[s] = (-tgt[s]) * tw[s]
            for s in range(3):
                Q[4 + s, 4 + s] = -opw[s]; q[4 + s] = opt[s] * (-opw[s])
            As.append(A); Bs.append(B); Qs.append(Q); qs.append(q); Rs.append(np.eye(2) * rc); x0 += ini
        ref.append(O.lq_solve(np.array(As), np.array(Bs), np.array(Qs), np.array(qs), np.array(Rs), np.array(x0), 3))
    return d, ref


@pytest.mark.parametrize("inactive_other", [False, True])
def test_pair_solver_arithmetic_equals_the_oracle(harness, inactive_other):
    rng = np.random.default_rng(11 + inactive_other)
    ng = 60
    d, ref = _games(rng, ng, inactive_other)
    inp = str(ng) + "\n" + "\n".join(repr(float(x)) for x in d.ravel())
    out = subprocess.run([harness], input=inp, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == ng
    for g, line in enumerate(out):
        a, b = [float.fromhex(t) for t in line.split()]
        assert a == ref[g][0] and b == ref[g][1], (g, a, ref[g][0], b, ref[g][1])
