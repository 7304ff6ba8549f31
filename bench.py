#!/usr/bin/env python3
"""bench.py — env-steps/s of the 4-agent Oval Fixed-LQNG race, 65 536 parallel envs per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
A "step" is one Unity FixedUpdate tick of EVERY env on the rank (K_A begin + K_B SolveLQR + K_C vehicle/engine kernels),
state resident in HBM.  For N > 1 the driver launches one rank per GPU with torch.distributed.run; envs shard
contiguously over ranks (no data-path collective), and the episode results are all-gathered over RCCL after the timed
region (the path's only exchange step).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 1056.0       # state in + state out at A = 4 (2 x 528 B)
ALGO_FLOP_PER_ENV_STEP = 330e3         # dense-equivalent fp64 flop (4 ego solves x 322 kflop / 4-tick cadence + physics)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6         # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


def cpu_baseline(num_agents, warmup, seed):
    """The CPU oracle (a line-by-line port of the reference C#, NOT the reference itself: no dotnet/Unity on the box)
    timed on the host cores with OpenMP over envs, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import hierarchicalkarting_amd as hk
    E, ticks = 512, 200
    b = hk.make_config(E, num_agents, jitter_seed=seed)
    o = O.OracleEnv(b)
    o.reset()
    o.step(warmup)
    t0 = time.perf_counter()
    o.step(ticks)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": E * ticks / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (C port of the reference C#), %d envs x %d ticks after %d warmup ticks, OpenMP over envs" % (E, ticks, warmup)}


FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA = the fp32 vector rate


def bench_rl(a, rank, local_rank, world, dist, torch, hk):
    """2v2 Oval, every agent LowMode RL (HighMode Fixed): per 2-tick decision the actor of the reference's
    HierarchicalAgent-Team-*scaledown* models (312 -> 256 x 3 Swish -> {mu, 3 logits}; random-init weights of that
    architecture) runs on device for all E x 4 agents.  One JSON line, roofline of policy_mlp_kernel against the f32 MFMA."""
    from hierarchicalkarting_amd import _lib
    from hierarchicalkarting_amd.policy import Policy
    E = a.envs_per_gpu
    A = 4
    env = hk.RacingEnv(hk.make_config(E, A, low_mode=[_lib.HK_LOW_RL] * A, jitter_seed=0x5EED0000, env_id_base=rank * E, device_id=local_rank))
    in_dim = env.obs_dim * 4
    p1 = Policy.random(in_dim, 256, 3, seed=101)
    p2 = Policy.random(in_dim, 256, 3, seed=202)
    env.attach_policy(p1, [0, 1], 2)
    env.attach_policy(p2, [2, 3], 2)
    env.reset()
    env.step(a.warmup)
    env.synchronize()

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        env.synchronize()

    env.prof_enable(True)
    env.prof_reset()
    barrier()
    t0 = time.perf_counter()
    env.step(a.steps)
    env.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = env.prof_read()
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        avg = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
        rows = E * 2                                   # rows per launch (one launch per team policy)
        flop_row = 2.0 * (in_dim * 256 + 2 * 256 * 256 + 4 * 256)
        ms = avg["policy_mlp_kernel"]
        ach = rows * flop_row / 1e12 / (ms * 1e-3) if ms > 0 else 0.0
        out = {"metric": "env-steps/sec (2v2 Oval, RL low-level on device)", "value": E * world * a.steps / dt, "unit": "env-steps/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "2v2 Oval, 4 agents LowMode RL / HighMode Fixed, %d envs per GPU, DecisionPeriod 2, one actor per team "
                                      "(%d -> 256 x 3 Swish -> mu + 3 logits, random-init weights of the reference architecture)" % (E, in_dim),
                          "envs_per_gpu": E, "agents": A},
               "roofline": {"bound": "mfma", "kernel": "policy_mlp_kernel", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms, "launches": prof["policy_mlp_kernel"][1],
                            "flop_per_launch": rows * flop_row, "kernel_avg_ms": avg,
                            "kernel_total_ms": {k: v[0] for k, v in prof.items()}}}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_mcts(a, rank, local_rank, world, dist, torch, hk):
    """BASELINE.json configs[2]: 4-agent Complex track, every agent MCTS high level + LQNG low level, 16 384 envs per GPU.
    The planner's wall-clock budget of the reference is an iteration budget here (--mcts-iterations per 100-tick replan)."""
    from hierarchicalkarting_amd import _lib
    rl = a.workload == "mctsrl"
    a8 = a.workload == "a8"
    E = a.envs_per_gpu if a.envs_per_gpu != 65536 else (131072 if a8 else (32768 if rl else 16384))
    A = 8 if a8 else 4
    low = [_lib.HK_LOW_RL if rl else _lib.HK_LOW_LQR] * A
    if a8:      # configs[4]: "mixed MCTS-RL vs MCTS-LQNG" = team 1 (agents 0-3) MCTS + RL actor, team 0 (agents 4-7) MCTS + LQNG
        low = [_lib.HK_LOW_RL] * 4 + [_lib.HK_LOW_LQR] * 4
    env = hk.RacingEnv(hk.make_config(E, A, track="oval" if rl else "complex", high_mode=[_lib.HK_HIGH_MCTS] * A, tree_search_depth=8,
                                      low_mode=low,
                                      mcts_iterations=a.mcts_iterations, jitter_seed=0x5EED0000, env_id_base=rank * E, device_id=local_rank))
    if rl:      # configs[3]: one 312 -> 256 x 3 actor per team (random-init weights of the reference architecture), DecisionPeriod 2
        from hierarchicalkarting_amd.policy import Policy
        env.attach_policy(Policy.random(env.obs_dim * 4, 256, 3, seed=101), [0, 1], 2)
        env.attach_policy(Policy.random(env.obs_dim * 4, 256, 3, seed=202), [2, 3], 2)
    if a8:      # one 504 -> 256 x 3 actor for the RL team
        from hierarchicalkarting_amd.policy import Policy
        env.attach_policy(Policy.random(env.obs_dim * 4, 256, 3, seed=303), [0, 1, 2, 3], 2)
    env.reset()
    env.step(a.warmup)
    env.synchronize()

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        env.synchronize()

    env.prof_enable(True)
    env.prof_reset()
    barrier()
    t0 = time.perf_counter()
    env.step(a.steps)
    env.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = env.prof_read()
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        m = env.mcts_state()
        out = {"metric": "env-steps/sec (8-agent Complex, mixed MCTS-RL vs MCTS-LQNG)" if a8 else
                         ("env-steps/sec (2v2 OvalDuos, MCTS-RL)" if rl else "env-steps/sec (4-agent Complex, MCTS-LQNG)"),
               "value": E * world * a.steps / dt, "unit": "env-steps/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": ("8-agent (4v4, synthetic: the reference has no 8-agent scene) Complex track, " if a8 else
                                       ("2v2 Oval, " if rl else "4-agent Complex track (41 sections), ")) +
                                      "MCTS high level (%d iterations per replan, depth 8, replan every 100 ticks) + %s low level, %d envs per GPU"
                                      % (a.mcts_iterations, "RL actor on device (504 -> 256 x 3, DecisionPeriod 2) for one team, LQNG for the other" if a8 else
                                         ("RL actor on device (312 -> 256 x 3 per team, DecisionPeriod 2)" if rl else "LQNG"), E),
                          "envs_per_gpu": E, "agents": A,
                          "searches_per_agent_mean": float(m["searches"].mean())},
               "kernel_total_ms": {k: v[0] for k, v in prof.items()},
               "note": "the planner kernel (mcts_search_kernel, one lane per search) is not bracketed by hk_prof: its share = wall time - the stages above"}
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 4 096 ticks per env, steady state = ticks 512-3 584 -> warm-up 512, timed 3 072
    ap.add_argument("--steps", type=int, default=3072)
    ap.add_argument("--warmup", type=int, default=512)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mcts-iterations", type=int, default=64)
    ap.add_argument("--workload", choices=("lqng", "rl", "mcts", "mctsrl", "a8"), default="lqng",
                    help="lqng: BASELINE.json configs[1] (the headline); rl: 2v2 Oval with the RL low-level actor on device (configs[3] shape); "
                         "mcts: 4-agent Complex track, MCTS-LQNG, 16 384 envs (configs[2]); "
                         "mctsrl: 2v2 OvalDuos, MCTS high level + RL low level on device, 32 768 envs per GPU (configs[3]); "
                         "a8: 8-agent Complex, mixed MCTS-RL vs MCTS-LQNG, 131 072 envs per GPU (configs[4])")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or os.environ.get("HK_BENCH_FORCE_DIST") == "1":      # (the flag: exercise the RCCL path with one rank on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libhk has no CPU fallback")

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd.parallel import gather_episode_results

    if a.workload == "rl":
        return bench_rl(a, rank, local_rank, world, dist, torch, hk)
    if a.workload in ("mcts", "mctsrl", "a8"):
        return bench_mcts(a, rank, local_rank, world, dist, torch, hk)
    E = a.envs_per_gpu
    seed = 0x5EED0000
    env = hk.RacingEnv(hk.make_config(E, a.agents, jitter_seed=seed, env_id_base=rank * E, device_id=local_rank))
    env.reset()
    env.step(a.warmup)
    env.synchronize()

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        env.synchronize()

    env.prof_enable(True)
    env.prof_reset()
    barrier()
    t0 = time.perf_counter()
    env.step(a.steps)
    env.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = env.prof_read()
    env.prof_enable(False)
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the path's one exchange step: all-gather of the episode results (RCCL over xGMI), off the timed region
    results = gather_episode_results(env, dist)
    total_envs = E * world
    value = total_envs * a.steps / dt

    if rank == 0:
        avg = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}           # ms per launch, HIP events
        dom = "env_run_kernel"                                                       # the fused tick kernel
        dom_ms = avg[dom]
        # one launch of the fused kernel advances the envs by a variable number of ticks (<= RUN_CAP = 32): the units
        # one launch processes = env-steps of the timed region / launches of the timed region
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * E * a.steps / max(prof[dom][1], 1)
        achieved = algo_bytes / 1e9 / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and E == 65536 and a.agents == 4:
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what the solves really were: players per game over the live egos at the end of the timed region
        st = env.agent_state()
        import numpy as np
        live = (st["flags"] & 4) != 0
        dx = st["px"][:, :, None] - st["px"][:, None, :]
        dz = st["pz"][:, :, None] - st["pz"][:, None, :]
        near = (np.sqrt(dx * dx + dz * dz) < 8.0).sum(axis=2)                        # players within 8 m incl. self (HKA:714)
        hist = np.bincount(near[live].ravel(), minlength=a.agents + 1)[1:]
        hist = (hist / max(hist.sum(), 1)).round(4).tolist()

        def lq_flop(N):                                                              # SURVEY §8 a1 dense flop formula
            n, m = 4 * N, 2 * N
            return 4 * (N * (4 * n ** 3 + 12 * n ** 2) + 2.0 / 3 * m ** 3 + 2 * m * m * (n + 1) + 2 * m * n)
        exec_flop = sum(h * lq_flop(i + 1) for i, h in enumerate(hist)) * a.agents / (4 if a.agents > 2 else 1) + 2000.0 * a.agents
        tick_ms = dt / a.steps * 1e3
        out = {
            "metric": "env-steps/sec (4-agent Oval, batch=65k)" if (a.agents == 4 and E == 65536) else "env-steps/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": tick_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d-agent Oval, Fixed-LQNG vs Fixed-LQNG (2v2), %d parallel envs per GPU, seeded start jitter, auto-reset"
                                   % (a.agents, E), "envs_per_gpu": E, "agents": a.agents, "sharding": "envs split contiguously over ranks",
                       "ticks": "warm-up %d then %d timed (BASELINE.md: steady state = ticks 512-3584)" % (a.warmup, a.steps),
                       "players_per_game_hist_N1..": hist,
                       "finished_episodes_seen": int((results["episode"] >= 0).any(axis=1).sum())},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": dom_ms, "launches": prof[dom][1],
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "whole_job": {"achieved": ALGO_BYTES_PER_ENV_STEP * E / 1e9 / (tick_ms * 1e-3),
                                       "frac": ALGO_BYTES_PER_ENV_STEP * E / 1e9 / (tick_ms * 1e-3) / HBM_PEAK_GBS,
                                       "note": "algorithmic bytes of the timed region / its wall time (all kernels + launch gaps)"},
                         "env_run_kernel_total_ms": prof[dom][0],
                         "kernel_avg_ms": avg,
                         "fp64_valu": {"dense_equivalent_flop_per_env_step_N4": ALGO_FLOP_PER_ENV_STEP,
                                       "executed_flop_per_env_step_at_measured_N": exec_flop,
                                       "achieved_tflops_dense_equivalent": ALGO_FLOP_PER_ENV_STEP * value / world / 1e12,
                                       "achieved_tflops_at_measured_N": exec_flop * value / world / 1e12,
                                       "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                       "frac_at_measured_N": exec_flop * value / world / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                       "note": "SURVEY §8d priced the path at N = 4 players per game; once the field spreads (> 8 m) ~99 % of the games are single-player"}},
        }
        if not a.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.agents, a.warmup, seed)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
