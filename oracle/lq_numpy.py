"""
oracle/lq_numpy.py — CPU ORACLE (test infrastructure): independent numpy mirror of the LQ Nash core.

Second, independently written restatement of components a1-a3:
  solve_feedback_lqr  <- AI/LQR/KartLQR.cs:17-128          (uses numpy matmul + numpy.linalg.solve, i.e. LAPACK getrf/getrs,
                                                             NOT the hand-written LU of hk_oracle_lq.c)
  bicycle_AB          <- AI/LQR/KartLQRDynamics.cs:40-62
  reach_avoid_cost    <- AI/LQR/KartLQRCosts.cs:57-140
It exists to (i) cross-check the C oracle (two restatements must agree to ~1e-12) and (ii) emit the golden vectors
under tests/golden/lq_*.json (python oracle/lq_numpy.py --emit).  The reference C# cannot run here (no dotnet), and it
ships no golden vectors, so these fixtures are the pin for this component.  Only tests/ may import this module.
"""
import json, os, sys
import numpy as np

XI, ZI, VI, HI = 0, 1, 2, 3          # AI/MPC/KartMPC.cs:15-18


def bicycle_AB(dt, initial):
    """KartLQRDynamics.cs:40-62"""
    A = np.eye(4)
    h, v = initial[HI], initial[VI]
    A[XI, VI] = np.cos(h) * dt
    A[ZI, VI] = np.sin(h) * dt
    A[XI, HI] = -np.sin(h) * dt * v
    A[ZI, HI] = np.cos(h) * dt * v
    B = np.zeros((4, 2))
    B[VI, 0] = dt
    B[HI, 1] = dt
    return A, B


def reach_avoid_cost(target, target_w, control_w, avoid_w, opp_target, opp_w):
    """KartLQRCosts.cs:57-140. avoid_w[2][M], opp_target[M][4], opp_w[M][3] -> Q (n,n), q (n), R (2,2)"""
    avoid_w = np.asarray(avoid_w, float).reshape(2, -1)
    M = avoid_w.shape[1]
    n = 4 + 4 * M
    Q = np.zeros((n, n))
    for s in (XI, ZI):
        total = 0.0
        for i in range(M):
            t = 4 + 4 * i + s
            w = avoid_w[s, i]
            Q[s, t] = w; Q[t, s] = w; Q[t, t] = -w
            total -= w
        Q[s, s] = total
    for k in range(4):
        Q[k, k] += target_w[k]
    opp_w = np.asarray(opp_w, float).reshape(M, 3)
    opp_target = np.asarray(opp_target, float).reshape(M, 4)
    for i in range(M):
        for k in range(3):
            Q[4 + 4 * i + k, 4 + 4 * i + k] = -opp_w[i, k]          # assignment (quirk Q4)
    q = np.zeros(n)
    q[:4] = -np.asarray(target, float) * np.asarray(target_w, float)
    for i in range(M):
        blk = opp_target[i].copy()
        blk[:3] = blk[:3] * -opp_w[i]
        q[4 + 4 * i: 8 + 4 * i] = blk
    R = np.eye(2) * control_w
    return Q, q, R


def solve_feedback_lqr(As, Bs_loc, Qs, qs, Rs, x0, horizon, trace=None):
    """KartLQR.cs:17-128.  As[N][4][4], Bs_loc[N][4][2], Qs[N][n][n], qs[N][n], Rs[N][2][2], x0[n]"""
    N = len(As); n = 4 * N; m = 2 * N
    A = np.zeros((n, n))
    Bs = []
    for i in range(N):
        A[4*i:4*i+4, 4*i:4*i+4] = As[i]
        B = np.zeros((n, 2)); B[4*i:4*i+4, :] = Bs_loc[i]; Bs.append(B)
    Zs = [np.array(Q, float) for Q in Qs]
    etas = [np.array(q, float) for q in qs]
    P = alpha = None
    for t in range(horizon, -1, -1):
        LHS = np.zeros((m, m))
        for i in range(N):                 # column block i, row block j (quirk Q1)
            for j in range(N):
                blk = Bs[i].T @ (Zs[i] @ Bs[j])
                if i == j:
                    blk = Rs[i] + blk
                LHS[2*j:2*j+2, 2*i:2*i+2] = blk
        RHSMat = np.vstack([Bs[i].T @ (Zs[i] @ A) for i in range(N)])
        RHSVec = np.concatenate([Bs[i].T @ etas[i] for i in range(N)])
        P = np.linalg.solve(LHS, RHSMat)
        alpha = np.linalg.solve(LHS, RHSVec)
        if trace is not None:
            trace.append((P.copy(), alpha.copy()))
        F = A - sum(Bs[k] @ P[2*k:2*k+2, :] for k in range(N))
        beta = -sum(Bs[k] @ alpha[2*k:2*k+2] for k in range(N))
        for i in range(N):
            Pi = P[2*i:2*i+2, :]
            Zs[i] = Qs[i] + Pi.T @ (Rs[i] @ Pi) + F.T @ (Zs[i] @ F)
            etas[i] = qs[i] + Pi.T @ (Rs[i] @ alpha[2*i:2*i+2]) + F.T @ (etas[i] + Zs[i] @ beta)   # new Z_i (Q2)
    return -P[0:2, :] @ np.asarray(x0, float) - alpha[0:2]


def random_game(rng, N):
    """A synthetic game with the structure SolveLQR produces (HKA:726-1198): bicycle dynamics about random states,
    reach-avoid costs with weights in the ranges the heuristics generate."""
    n = 4 * N
    inits = []
    for _ in range(N):
        inits.append([rng.uniform(-50, 25), rng.uniform(-50, 65), rng.uniform(0, 15), rng.uniform(0, 2*np.pi)])
    near = max(N - 1, 1)
    As, Bs, Qs, qs, Rs = [], [], [], [], []
    for k in range(N):
        A, B = bicycle_AB(0.02, inits[k]); As.append(A); Bs.append(B)
        tgt = [inits[k][0] + rng.uniform(-10, 10), inits[k][1] + rng.uniform(-10, 10), 15.0, inits[k][3] + rng.uniform(-0.6, 0.6)]
        v = inits[k][2]
        if v <= 5:
            tw = [near*0.93, near*0.93, near*-2.0, 2.5*near]; tgt[2] = 0.0
        else:
            tw = [near*0.93/max(1, v), near*0.93/max(1, v), near*5e-4, 2.5*near]
        M = N - 1
        aw = np.zeros((2, M)); ot = np.zeros((M, 4)); ow = np.zeros((M, 3))
        for j in range(M):
            d = rng.uniform(1.0, 8.0)
            w = 1.0/(d**1.5 * rng.choice([0.55, 1.7, 0.45, 1.3])/near)
            aw[:, j] = w
            ot[j, :3] = [rng.uniform(-50, 25), rng.uniform(-50, 65), 15.0]
            ow[j] = [0.1/(max(1, v)*near), 0.1/(max(1, v)*near), 0.08/near]
        Q, q, R = reach_avoid_cost(tgt, tw, 0.135 if N > 2 else 0.115, aw, ot, ow)
        Qs.append(Q); qs.append(q); Rs.append(R)
    x0 = np.concatenate(inits)
    return As, Bs, Qs, qs, Rs, x0


def emit(outdir):
    os.makedirs(outdir, exist_ok=True)
    for N in (1, 2, 3, 4, 8):
        rng = np.random.default_rng(1000 + N)
        cases = []
        for c in range(6):
            As, Bs, Qs, qs, Rs, x0 = random_game(rng, N)
            tr = []
            u0 = solve_feedback_lqr(As, Bs, Qs, qs, Rs, x0, 3, trace=tr)
            cases.append({
                "N": N, "horizon": 3,
                "A": np.array(As).tolist(), "B": np.array(Bs).tolist(), "Q": np.array(Qs).tolist(),
                "q": np.array(qs).tolist(), "R": np.array(Rs).tolist(), "x0": x0.tolist(),
                "u0": u0.tolist(),
                "P_last": tr[-1][0].tolist(), "alpha_last": tr[-1][1].tolist(),
                "P_first": tr[0][0].tolist(), "alpha_first": tr[0][1].tolist(),
            })
        with open(os.path.join(outdir, "lq_N%d.json" % N), "w") as f:
            json.dump({"generator": "oracle/lq_numpy.py --emit (numpy %s)" % np.__version__, "cases": cases}, f)
        print("wrote", N, len(cases))


if __name__ == "__main__":
    if "--emit" in sys.argv:
        emit(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
