#!/usr/bin/env python3
"""bench.py — env-steps/s of the 4-agent Oval Fixed-LQNG race, 65 536 parallel envs per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

A "step" is one Unity FixedUpdate tick of EVERY env on the rank (hk_step: rounds of the fused tick kernel + the queued
multi-player LQ solves), state resident in HBM.  Protocol of the default workload (the metric is the
STEADY STATE of the race, ticks 512..3584):
    reset -> PRE-ROLL to tick 512 (untimed set-up, --preroll) -> W warm-up ticks (untimed) -> EXACTLY K timed ticks,
bracketed by barrier + synchronize on both sides, max over ranks.  `config.ticks` states the tick range that was timed.
After the timed region the same process also measures, as labelled secondary fields, the full BASELINE protocol
(ticks 512..3584 of a fresh race) and the race start (ticks 0..512: everyone within 8 m, 4-player games).

--gpus N with N > 1: when WORLD_SIZE is not set, THIS process starts N ranks (python -m torch.distributed.run, one per GPU)
as a child BEFORE anything touches the GPU, relays the child's JSON line and exits with its code; it fails cleanly when the
box has fewer than N GPUs.  Under torch.distributed.run (WORLD_SIZE set) it is one rank: envs shard contiguously over ranks
(no data-path collective), the episode results are all-gathered over RCCL after the timed region (the path's only exchange
step), rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 1056.0       # state in + state out at A = 4 (2 x 528 B)
ALGO_FLOP_PER_ENV_STEP = 330e3         # dense-equivalent fp64 flop (4 ego solves x 322 kflop / 4-tick cadence + physics)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6         # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA = the fp32 vector rate
STEADY_TICK = 512                      # steady state = ticks 512 .. 3584
# HBM bytes and SQ counters cannot be read in-process: they come from the separate rocprofv3 --pmc passes of THIS command that
# tools/pmc_summary.py folded into this file (committed with the profile it belongs to; `commit` / `command` inside say which)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
# the launches of a solve cadence in the steady state of the profiled run: tick + B1 (which solves the multi-player games in-wave).  lqn_spread_kernel runs in a fraction of
# the rounds (while the games meter reads "medium") and enters weighted by its share of launches; lqn_round_kernel is the race start's solver (the first 384 ticks of the
# profiled run) and is not part of the steady state
PMC_CADENCE_KERNELS = ("env_run_kernel", "env_b1_kernel", "lqn_spread_kernel")


def _cadence_weight(d, k):
    """launches of kernel k per tick launch in the profiled run (1 for the tick and B1 kernels)"""
    base = (d.get("env_run_kernel") or {}).get("launches_seen") or 0
    n = (d.get(k) or {}).get("launches_seen") or 0
    return min(1.0, n / base) if base else 0.0


def pmc_fields(kernel, env_steps_per_launch=None):
    """-> (traffic bytes per launch or None, provenance dict, binding dict) for `kernel` from the committed PMC summary; the bytes are
    scaled to this run's env-steps per launch (the fused tick kernel advances a variable number of ticks per launch)"""
    try:
        d = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None, {"note": "no PMC summary committed for this round"}, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    if d.get("sources_sha16") != source_hash():
        # the kernels changed after the counters were taken: no stale bytes in the line
        return None, {"file": os.path.relpath(PMC_SUMMARY, ROOT), "stale": True,
                      "note": "the committed PMC summary was measured on other kernel sources (sources_sha16 %s, this tree %s)" % (d.get("sources_sha16"), source_hash())}, None
    k = d.get(kernel) or {}
    prov = {"file": os.path.relpath(PMC_SUMMARY, ROOT), "commit": d.get("commit"), "command": d.get("command"), "sources_sha16": d.get("sources_sha16"),
            "method": "separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE; " + str(d.get("note", ""))}
    sq = (k.get("sq") or {}).get("derived")
    binding = None
    if sq:
        # (the PMC passes run every launch ALONE on the GPU — rocprofv3 serialises dispatches while it collects counters — i.e. a half-batch launch with
        # its two waves per SIMD; in the product schedule two streams' launches share the SIMDs, up to `waves_per_simd` resident: see valu_port_use_over_wall,
        # which main() adds from this run's own wall time)
        alone = 2.0
        binding = {"resource": "neither roof: vector-instruction issue and the waits / dependent-launch gaps the resident waves do not cover (profiles/r05_a_backend_flags.txt, r06_d_*, DESIGN.md section 2)",
                   "wave_issuing_valu_frac": sq.get("wave_issuing_valu_frac"), "wave_issuing_any_frac": sq.get("wave_issuing_any_frac"),
                   "wave_waiting_frac": sq.get("wave_waiting_frac"), "waves_per_simd_limit": d.get("waves_per_simd", 2),
                   "simd_valu_busy_frac_launch_alone": min(1.0, alone * (sq.get("wave_issuing_valu_frac") or 0.0)),
                   "valu_lanes_active_of_64": sq.get("valu_lanes_active_of_64"), "valu_insts_per_launch": sq.get("valu_insts_per_launch"),
                   "source": "SQ counters of the same PMC summary (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave; a launch alone on the GPU = 2 048 waves on 1 024 SIMDs)"}
    traffic = k.get("hbm_bytes_per_launch")
    if traffic is not None and env_steps_per_launch and d.get("env_steps_per_launch"):
        prov["measured_bytes_per_env_step"] = traffic / d["env_steps_per_launch"]
        traffic = traffic * env_steps_per_launch / d["env_steps_per_launch"]
    return traffic, prov, binding


def valu_port_use(tick_launches, wall_s):
    """Share of the SIMDs' vector-issue slots the timed region used: (vector instructions of a tick + B1 + solver launch set of the committed PMC summary,
    four cycles each on a SIMD) x this run's launch sets / (1 024 SIMDs x this run's wall time at the peak clock).  An instruction count does not depend on
    what else is resident, so this holds for the two-stream schedule although the counters were taken launch by launch."""
    try:
        d = json.load(open(PMC_SUMMARY))
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import source_hash
        if d.get("sources_sha16") != source_hash():
            return None          # the kernels changed after the counters were taken: no stale instruction counts in the line
        quad = sum(float(d[k]["sq"]["SQ_ACTIVE_INST_VALU"]) * _cadence_weight(d, k) for k in PMC_CADENCE_KERNELS if k in d and "sq" in d[k])
    except (OSError, ValueError, KeyError, TypeError):
        return None
    if wall_s <= 0:
        return None
    simds, clock_hz = 1024, 2.4e9
    return {"frac": quad * 4.0 * tick_launches / (simds * clock_hz * wall_s), "valu_busy_quad_cycles_per_launch_set": quad, "simds": simds, "clock_ghz": clock_hz / 1e9,
            "note": "SQ_ACTIVE_INST_VALU (units of four cycles) of tick + B1 launch (+ the spread solver's launch, weighted by its share of rounds), x launch sets of the timed region, / (SIMDs x wall cycles at the peak clock); the rest of the slots: waits no resident wave covers, the gaps between the three dependent launches of a round, ramp and tail of every launch"}


def cadence_traffic(env_steps_per_tick_launch):
    """HBM bytes of one solve cadence of the fission schedule = tick launch + B1 launch + solver launch of the committed PMC summary, scaled to this
    run's env-steps per tick launch (a reader of `traffic` alone sees the tick kernel only: the B1 kernel moves about as much)"""
    try:
        d = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import source_hash
    if d.get("sources_sha16") != source_hash() or not d.get("env_steps_per_launch"):
        return None
    parts = {k: (d[k].get("hbm_bytes_per_launch") or 0.0) * _cadence_weight(d, k) for k in PMC_CADENCE_KERNELS if k in d}
    if parts["env_run_kernel"] is None:
        return None
    scale = env_steps_per_tick_launch / d["env_steps_per_launch"]
    tot = sum(v for v in parts.values() if v) * scale
    return {"bytes_per_cadence_launch_set": tot, "bytes_per_env_step": tot / env_steps_per_tick_launch,
            "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "ratio_to_algorithmic": tot / env_steps_per_tick_launch / ALGO_BYTES_PER_ENV_STEP,
            "parts_bytes_per_launch": {k: (v * scale if v else None) for k, v in parts.items()}}


def lq_flop(N, sweeps=4):
    """dense flop of one KartLQR.solveFeedbackLQR call with N players (SURVEY §8 a1 formula; horizon 3 = 4 sweeps)"""
    n, m = 4 * N, 2 * N
    return sweeps * (N * (4 * n ** 3 + 12 * n ** 2) + 2.0 / 3 * m ** 3 + 2 * m * m * (n + 1) + 2 * m * n)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N > 1 without WORLD_SIZE -> N ranks as a child process group (nothing here touches the GPU)
# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(a, argv):
    selftest = a.selftest_launcher
    if not selftest:
        import torch                                   # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < (1 if a.same_device else a.gpus):
            print("bench.py: --gpus %d needs %d GPUs, this box has %d" % (a.gpus, a.gpus, have), file=sys.stderr)
            return 2
    port = int(os.environ.get("MASTER_PORT", 29500 + (os.getpid() % 2000)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print("bench.py: the %d-rank child failed (exit code %d)" % (a.gpus, p.returncode), file=sys.stderr)
        return p.returncode or 1
    print(line, flush=True)
    return 0


class Dist:
    """the rank's view of the job: barrier, max-over-ranks of a time, the result gather (RCCL on GPUs, gloo in the CPU selftest)"""

    def __init__(self, backend):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        import torch
        self.torch = torch
        if self.world > 1 or os.environ.get("HK_BENCH_FORCE_DIST") == "1":      # (the flag: exercise the RCCL path with one rank on a 1-GPU box)
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist
        self.dev = "cuda" if backend == "nccl" else "cpu"

    def barrier(self, env=None):
        if self.dist:
            self.dist.barrier()
        if self.backend == "nccl" or self.torch.cuda.is_initialized():
            self.torch.cuda.synchronize()
        if env is not None:
            env.synchronize()

    def sync_device(self):
        if self.backend == "nccl" or self.torch.cuda.is_initialized():
            self.torch.cuda.synchronize()

    def all_times(self, dt):
        """every rank's value of dt, rank order"""
        if not self.dist:
            return [dt]
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros(1, dtype=self.torch.float64, device=self.dev) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [float(x.item()) for x in out]

    def max_time(self, dt):
        if not self.dist:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_ticks(D, env, ticks, per_rank=None):
    """EXACTLY `ticks` ticks of every env on every rank -> seconds = the MAX over ranks of each rank's own span.  Every rank starts from a common
    barrier + synchronize; its span ends when ITS work is done (hk_synchronize: completion guard verified, stream drained; then the device-wide
    synchronize of the contract); the closing barrier comes AFTER the clock is read (round 6: inside the span it put an RCCL all-reduce and two device
    synchronisations into a window that lasts under a millisecond — at N > 1 the curve would have measured barrier latency, not the path).
    per_rank: a list that receives every rank's span in ms (world > 1)."""
    D.barrier(env)
    t0 = time.perf_counter()
    env.step(ticks)
    env.synchronize()
    D.sync_device()
    dt = time.perf_counter() - t0
    D.barrier(env)
    if per_rank is not None:
        per_rank[:] = [t * 1e3 for t in D.all_times(dt)]
    return D.max_time(dt)


def selftest_launcher(a):
    """CPU check of the N-rank plumbing (tests/test_bench_launcher.py): rendezvous, shard ranges that do not divide,
    barrier, max-over-ranks, the padded all-gather, one JSON line from rank 0.  No libhk compute, no GPU."""
    import numpy as np
    D = Dist("gloo")
    if D.world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, D.world))
    from hierarchicalkarting_amd.parallel import shard_range, gather_episode_results
    from hierarchicalkarting_amd.env import RESULT_DT
    total = a.envs_per_gpu * D.world + 1                     # +1: the last rank's shard is shorter than the others
    lo, hi = shard_range(total, D.rank, D.world)
    local = np.zeros((hi - lo, 2), RESULT_DT)
    local["time_steps"] = np.arange(lo, hi, dtype=np.int32)[:, None] * 2 + np.arange(2, dtype=np.int32)[None, :]
    D.barrier()
    dt = D.max_time(0.001 * (D.rank + 1))
    spans = D.all_times(0.001 * (D.rank + 1))
    allres = gather_episode_results(local, D.dist)
    ok = allres.shape == (total, 2) and bool((allres["time_steps"] == np.arange(total, dtype=np.int32)[:, None] * 2 + np.arange(2)[None, :]).all())
    if D.rank == 0:
        print(json.dumps({"metric": "launcher selftest", "n_gpus": D.world, "gathered_envs": int(allres.shape[0]),
                          "gather_ok": ok, "max_time": dt, "per_rank_ms": [t * 1e3 for t in spans], "ranks_seen": int(D.dist.get_world_size()), "selftest": True}), flush=True)
    D.close()
    return 0 if ok else 1


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed beside the GPU path on rank 0 at N = 1)
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(num_agents, seed, start_tick):
    """the CPU oracle (a line-by-line C port of the reference C#, NOT the reference itself: no
    dotnet / Unity on either box) at E = 4 096 on the same inputs, from the same tick the GPU window starts at; timed on
    all host cores (OpenMP over envs, >= 16 envs per thread) and on ONE thread.  Bounded: ~10 s each."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import hierarchicalkarting_amd as hk
    E = 4096
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a container can see every host core and still be granted a fraction of them (cgroup CPU quota): threads beyond the grant
    # only fight each other (a GPU box of this pool shows 256 cores and grants ~16)
    granted = cores
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    granted = min(granted, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    granted = min(granted, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    threads = max(1, min(granted, E // 16))
    O.set_threads(threads)
    o = O.OracleEnv(hk.make_config(E, num_agents, jitter_seed=seed))
    o.reset()
    o.step(start_tick)                                       # untimed: the same pre-roll as the GPU run

    def sample(budget_s, chunk, max_ticks):
        done, t0 = 0, time.perf_counter()
        while done < max_ticks and (time.perf_counter() - t0) < budget_s:
            o.step(chunk)
            done += chunk
        return done, time.perf_counter() - t0
    ticks_all, dt_all = sample(10.0, 16, 2048)
    O.set_threads(1)
    ticks_one, dt_one = sample(10.0, 2, 256)
    O.set_threads(threads)
    return {"value": E * ticks_all / dt_all, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "CPU oracle (C port of the reference C#; the reference itself cannot run here), E = %d envs, ticks %d..%d "
                      "of the same workload, OpenMP over envs on %d threads (%d host cores visible, %d granted to this container)"
                      % (E, start_tick, start_tick + ticks_all, threads, cores, granted),
            "single_thread": {"value": E * ticks_one / dt_one, "unit": "env-steps/s", "cores": 1,
                              "sample": "same envs, the next %d ticks on 1 thread" % ticks_one}}


# ----------------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------------
def bench_rl(a, D, hk):
    """2v2 Oval, every agent LowMode RL (HighMode Fixed): per 2-tick decision the actor of the reference's
    HierarchicalAgent-Team-*scaledown* models (312 -> 256 x 3 Swish -> {mu, 3 logits}; random-init weights of that
    architecture) runs on device for all E x 4 agents.  One JSON line, roofline of policy_mlp_kernel against the f32 MFMA."""
    from hierarchicalkarting_amd import _lib
    from hierarchicalkarting_amd.policy import Policy
    E = a.envs_per_gpu
    A = 4
    env = hk.RacingEnv(hk.make_config(E, A, low_mode=[_lib.HK_LOW_RL] * A, jitter_seed=0x5EED0000, env_id_base=D.rank * E, device_id=D.local_rank))
    in_dim = env.obs_dim * 4
    env.attach_policy(Policy.random(in_dim, 256, 3, seed=101), [0, 1], 2)
    env.attach_policy(Policy.random(in_dim, 256, 3, seed=202), [2, 3], 2)
    env.reset()
    env.step(a.warmup)
    env.synchronize()
    env.prof_enable(True)
    env.prof_reset()
    dt = timed_ticks(D, env, a.steps)
    prof = env.prof_read()
    if D.rank == 0:
        avg = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
        rows = E * 2                                   # rows per launch (one launch per team policy)
        flop_row = 2.0 * (in_dim * 256 + 2 * 256 * 256 + 4 * 256)
        ms = avg["policy_mlp_kernel"]
        ach = rows * flop_row / 1e12 / (ms * 1e-3) if ms > 0 else 0.0
        out = {"metric": "env-steps/sec (2v2 Oval, RL low-level on device)", "value": E * D.world * a.steps / dt, "unit": "env-steps/s",
               "n_gpus": D.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "2v2 Oval, 4 agents LowMode RL / HighMode Fixed, %d envs per GPU, DecisionPeriod 2, one actor per team "
                                      "(%d -> 256 x 3 Swish -> mu + 3 logits, random-init weights of the reference architecture)" % (E, in_dim),
                          "envs_per_gpu": E, "agents": A},
               "roofline": {"bound": "mfma", "kernel": "policy_mlp_kernel", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms, "launches": prof["policy_mlp_kernel"][1],
                            "flop_per_launch": rows * flop_row, "kernel_avg_ms": avg,
                            "kernel_total_ms": {k: v[0] for k, v in prof.items()}}}
        print(json.dumps(out), flush=True)
    D.close()


def bench_mcts(a, D, hk):
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
                                      mcts_iterations=a.mcts_iterations, jitter_seed=0x5EED0000, env_id_base=D.rank * E, device_id=D.local_rank))
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
    env.prof_enable(True)
    env.prof_reset()
    dt = timed_ticks(D, env, a.steps)
    prof = env.prof_read()
    if D.rank == 0:
        m = env.mcts_state()
        out = {"metric": "env-steps/sec (8-agent Complex, mixed MCTS-RL vs MCTS-LQNG)" if a8 else
                         ("env-steps/sec (2v2 OvalDuos, MCTS-RL)" if rl else "env-steps/sec (4-agent Complex, MCTS-LQNG)"),
               "value": E * D.world * a.steps / dt, "unit": "env-steps/s",
               "n_gpus": D.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": ("8-agent (4v4, synthetic: the reference has no 8-agent scene) Complex track, " if a8 else
                                       ("2v2 Oval, " if rl else "4-agent Complex track (41 sections), ")) +
                                      "MCTS high level (%d iterations per replan, depth 8, replan every 100 ticks) + %s low level, %d envs per GPU"
                                      % (a.mcts_iterations, "RL actor on device (504 -> 256 x 3, DecisionPeriod 2) for one team, LQNG for the other" if a8 else
                                         ("RL actor on device (312 -> 256 x 3 per team, DecisionPeriod 2)" if rl else "LQNG"), E),
                          "envs_per_gpu": E, "agents": A,
                          "searches_per_agent_mean": float(m["searches"].mean())},
               "kernel_total_ms": {k: v[0] for k, v in prof.items()},
               "note": "the planner kernel (mcts_search_kernel) is not bracketed by hk_prof: its share = wall time - the stages above"}
        print(json.dumps(out), flush=True)
    D.close()


def bench_lqbatch(a, D, hk):
    """K1 alone (SURVEY §7 minimum slice): hk_lq_solve_batch_device on 262 144 random N-player games resident in HBM, N = 2 and 4;
    solves/s, algorithmic GB/s and the fraction of the fp64 vector peak (dense flop formula of SURVEY §8 a1)."""
    import ctypes as C
    import numpy as np
    from hierarchicalkarting_amd import _lib
    torch = D.torch

    def random_game(rng, N):
        """inputs with the structure SolveLQR hands to solveFeedbackLQR (HKA:726-1198): linearised bicycles about random kart
        states, reach-avoid costs with weights in the ranges the heuristics generate.  (A, B, Q, q, R, x0)"""
        n = 4 * N
        near = max(N - 1, 1)
        dt = 0.02
        As, Bs, Qs, qs, Rs, x0 = [], [], [], [], [], []
        for k in range(N):
            x, z, v, th = rng.uniform(-50, 25), rng.uniform(-50, 65), rng.uniform(0, 15), rng.uniform(0, 2 * np.pi)
            x0 += [x, z, v, th]
            Ak = np.eye(4)
            Ak[0, 2], Ak[1, 2], Ak[0, 3], Ak[1, 3] = np.cos(th) * dt, np.sin(th) * dt, -np.sin(th) * dt * v, np.cos(th) * dt * v
            Bk = np.zeros((4, 2)); Bk[2, 0] = dt; Bk[3, 1] = dt
            tw = np.array([near * 0.93 / max(1, v), near * 0.93 / max(1, v), near * 5e-4, 2.5 * near])
            tgt = np.array([x + rng.uniform(-10, 10), z + rng.uniform(-10, 10), 15.0, th + rng.uniform(-0.6, 0.6)])
            Q = np.zeros((n, n)); q = np.zeros(n)
            for j in range(1, N):                              # avoid terms on (x, z) pairs, then the opponent-target diagonal
                w = 1.0 / (rng.uniform(1.0, 8.0) ** 1.5 * 1.3 / near)
                for s_ in range(2):
                    Q[s_, s_] -= w; Q[s_, 4 * j + s_] = w; Q[4 * j + s_, s_] = w
                ow = np.array([0.1 / (max(1, v) * near), 0.1 / (max(1, v) * near), 0.08 / near])
                ot = np.array([rng.uniform(-50, 25), rng.uniform(-50, 65), 15.0])
                for s_ in range(3):
                    Q[4 * j + s_, 4 * j + s_] = -ow[s_]; q[4 * j + s_] = -ow[s_] * ot[s_]
            for s_ in range(4):
                Q[s_, s_] += tw[s_]; q[s_] = -tgt[s_] * tw[s_]
            As.append(Ak); Bs.append(Bk); Qs.append(Q); qs.append(q); Rs.append(np.eye(2) * (0.135 if N > 2 else 0.115))
        return As, Bs, Qs, qs, Rs, np.array(x0)

    L = _lib.load()
    h = C.c_void_p()
    _lib.check(L.hk_create(None, C.byref(h)), None)
    batch = a.lq_batch
    rows = {}
    for N in (2, 4):
        rng = np.random.default_rng(1000 + N)
        base = [random_game(rng, N) for _ in range(256)]
        arrs = [np.array([g[k] for g in base]) for k in range(6)]
        dev = []
        for x in arrs:                                     # tile the 256 distinct games to the batch, on device
            t = torch.from_numpy(np.ascontiguousarray(x, np.float64)).cuda()
            reps = [batch // 256] + [1] * (t.dim() - 1)
            dev.append(t.repeat(*reps).contiguous())
        u0 = torch.zeros(batch, 2, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ptr = lambda t: C.c_void_p(t.data_ptr())

        def run():
            _lib.check(L.hk_lq_solve_batch_device(h, batch, N, ptr(dev[0]), ptr(dev[1]), ptr(dev[2]), ptr(dev[3]), ptr(dev[4]), ptr(dev[5]), 3, ptr(u0), None), h)
        for _ in range(a.warmup if a.warmup < 50 else 5):
            run()
        _lib.check(L.hk_synchronize(h), h)
        _lib.check(L.hk_prof_enable(h, 1), h)
        _lib.check(L.hk_prof_reset(h), h)
        iters = a.steps if a.steps < 500 else 20
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        _lib.check(L.hk_synchronize(h), h)
        dt = time.perf_counter() - t0
        ms = (C.c_double * _lib.HK_PROF_STAGES)()
        n = (C.c_int64 * _lib.HK_PROF_STAGES)()
        _lib.check(L.hk_prof_read(h, ms, n), h)
        k_ms = ms[2] / max(n[2], 1)
        nn = 4 * N
        bytes_game = 8.0 * (N * 16 + N * 8 + N * nn * nn + N * nn + N * 4 + nn + 2)
        tf = batch * lq_flop(N) / 1e12 / (k_ms * 1e-3)
        rows["N%d" % N] = {"solves_per_s_wall": batch * iters / dt, "solves_per_s_kernel": batch / (k_ms * 1e-3), "kernel_ms": k_ms,
                           "dense_flop_per_solve": lq_flop(N), "achieved_tflops": tf, "frac_fp64_vector_peak": tf / FP64_VECTOR_PEAK_TFLOPS,
                           "algorithmic_bytes_per_solve": bytes_game, "achieved_gbs": batch * bytes_game / 1e9 / (k_ms * 1e-3),
                           "frac_hbm_peak": batch * bytes_game / 1e9 / (k_ms * 1e-3) / HBM_PEAK_GBS}
    r4 = rows["N4"]
    out = {"metric": "LQ Nash solves/sec (hk_lq_solve_batch, N = 4 players, batch 262 144)", "value": r4["solves_per_s_wall"], "unit": "solves/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r4["kernel_ms"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "KartLQR.solveFeedbackLQR batched: %d random games per launch (256 distinct, tiled), horizon 3, dense Q / q / R inputs" % batch},
           "roofline": {"bound": "fp64_valu", "kernel": "lq_batch_kernel<4>", "achieved": r4["achieved_tflops"], "peak": FP64_VECTOR_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": r4["frac_fp64_vector_peak"], "traffic": None},
           "per_N": rows}
    print(json.dumps(out), flush=True)
    L.hk_destroy(h)
    D.close()


def bench_lqng(a, D, hk):
    """the headline: BASELINE.json configs[1]"""
    import numpy as np
    from hierarchicalkarting_amd.parallel import gather_episode_results
    E, A = a.envs_per_gpu, a.agents
    seed = 0x5EED0000
    env = hk.RacingEnv(hk.make_config(E, A, jitter_seed=seed, env_id_base=D.rank * E, device_id=D.local_rank))
    env.reset()
    if a.preroll > 0:
        env.step(a.preroll)                              # untimed set-up: to the start of the steady state
    if a.warmup > 0:
        env.step(a.warmup)
    env.synchronize()
    per_rank_ms = []
    dt = timed_ticks(D, env, a.steps, per_rank_ms)       # THE timed region: hk_prof off (its event records cost ~35 us per call)
    sched = env.schedule_info()                          # what the library ran for that call (hk_schedule_info)
    st = env.agent_state() if D.rank == 0 else None
    # the same window again, back to back (up to 15 times while the race stays inside the protocol's steady stretch): how much of `value` is the
    # noise of one short sample.  `value` stays the ONE window above.
    n_rep = max(0, min(15, (STEADY_TICK + 3072 - (a.preroll + a.warmup + a.steps)) // max(a.steps, 1)))
    rep_dt = [timed_ticks(D, env, a.steps) for _ in range(n_rep)]
    # the roofline's stage times come from a second, identical pass (same reset, pre-roll, warm-up, ticks) with hk_prof on; not `value`
    env.reset()
    if a.preroll > 0:
        env.step(a.preroll)
    if a.warmup > 0:
        env.step(a.warmup)
    env.synchronize()
    env.prof_enable(True)
    env.prof_reset()
    dt_prof = timed_ticks(D, env, a.steps)
    prof = env.prof_read()
    games = env.prof_games()
    env.prof_enable(False)
    st2 = env.agent_state() if D.rank == 0 else None
    if D.rank == 0:                                      # the two passes ran the same race: bit-identical karts (the TelemetryViewer fields outlive a reset)
        for f in ("px", "pz", "yaw", "vx", "vz", "wy", "section_index", "flags"):
            assert np.array_equal(st[f], st2[f]), "the profiled pass did not reproduce the timed pass (%s)" % f
    # the path's one exchange step: all-gather of the episode results (RCCL over xGMI), off the timed region
    D.barrier(env)
    tg0 = time.perf_counter()
    results = gather_episode_results(env, D.dist)
    gather_ms = (time.perf_counter() - tg0) * 1e3
    value = E * D.world * a.steps / dt
    t_first = a.preroll + a.warmup

    # secondary, labelled: the full BASELINE protocol window and the race start, each on a fresh race
    secondary = {}
    if not a.no_secondary:
        env.reset()
        env.prof_enable(True); env.prof_reset()
        dt_rs = timed_ticks(D, env, STEADY_TICK)
        prof_rs = env.prof_read()
        games_rs = env.prof_games()
        env.prof_reset()
        dt_full = timed_ticks(D, env, 3072)
        prof_full = env.prof_read()
        games_full = env.prof_games()
        env.prof_enable(False)
        # host-driven mode: a Unity host calls hk_step(1) per FixedUpdate, an ML-Agents loop hk_step(2) between two
        # decisions.  From the steady state (tick 512) of a fresh race: 256 one-tick calls, then 128 two-tick calls, no getter in between.
        env.reset()
        env.step(STEADY_TICK)

        def short_calls(n_calls, n):
            D.barrier(env)
            t0 = time.perf_counter()
            for _ in range(n_calls):
                env.step(n)
            env.synchronize()
            D.barrier(env)
            return D.max_time(time.perf_counter() - t0)
        dt_1 = short_calls(256, 1)
        dt_2 = short_calls(128, 2)
        dt_20 = short_calls(16, 20)
        # the same protocol window with the batch on ONE stream (HK_SPLIT=0, read in hk_create): what the default schedule — two halves on
        # two streams, a half's solver launch hidden behind the other half's tick launch — buys; and the per-launch roofline of the tick
        # kernel with the GPU to itself
        env.close()                                       # (its streams go back first)
        old_split = os.environ.get("HK_SPLIT")
        os.environ["HK_SPLIT"] = "0"
        try:
            env2 = hk.RacingEnv(hk.make_config(E, A, jitter_seed=seed, env_id_base=D.rank * E, device_id=D.local_rank))
        finally:
            if old_split is None:
                del os.environ["HK_SPLIT"]
            else:
                os.environ["HK_SPLIT"] = old_split
        env2.reset(); env2.step(STEADY_TICK); env2.synchronize()
        env2.prof_enable(True); env2.prof_reset()
        dt_one = timed_ticks(D, env2, 3072)
        prof_one = env2.prof_read()
        env2.prof_enable(False)
        del env2
        one_run = prof_one["env_run_kernel"]
        one_ms = one_run[0] / max(one_run[1], 1)
        one_algo = ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E * 3072 / max(one_run[1], 1)
        secondary = {
            "host_driven": {"hk_step(1)_x256_from_tick_512": {"value": E * D.world * 256 / dt_1, "unit": "env-steps/s", "us_per_call": dt_1 / 256 * 1e6},
                            "hk_step(2)_x128_from_tick_768": {"value": E * D.world * 256 / dt_2, "unit": "env-steps/s", "us_per_call": dt_2 / 128 * 1e6},
                            "hk_step(20)_x16_from_tick_1024": {"value": E * D.world * 320 / dt_20, "unit": "env-steps/s", "us_per_call": dt_20 / 16 * 1e6}},
            "one_stream_ticks_512_3584": {"value": E * D.world * 3072 / dt_one, "unit": "env-steps/s",
                                          "env_run_kernel": {"launches": one_run[1], "avg_launch_ms": one_ms, "algorithmic_bytes_per_launch": one_algo,
                                                             "achieved_gbs": one_algo / 1e9 / (one_ms * 1e-3) if one_ms > 0 else 0.0,
                                                             "frac_hbm_peak": (one_algo / 1e9 / (one_ms * 1e-3) / HBM_PEAK_GBS) if one_ms > 0 else 0.0},
                                          "note": "HK_SPLIT=0 (not the default): the whole batch on one stream, the tick kernel alone on the GPU — its own per-launch roofline"},
            "baseline_protocol_ticks_512_3584": {"value": E * D.world * 3072 / dt_full, "unit": "env-steps/s", "seconds": dt_full,
                                                 "kernel_total_ms": {k: v[0] for k, v in prof_full.items() if v[1]},
                                                 "launches": {k: v[1] for k, v in prof_full.items() if v[1]},
                                                 "multi_player_games": {str(k): v for k, v in games_full.items() if v}},
            "race_start_ticks_0_512": {"value": E * D.world * STEADY_TICK / dt_rs, "unit": "env-steps/s", "seconds": dt_rs,
                                       "kernel_total_ms": {k: v[0] for k, v in prof_rs.items() if v[1]},
                                       "launches": {k: v[1] for k, v in prof_rs.items() if v[1]},
                                       "multi_player_games": {str(k): v for k, v in games_rs.items() if v},
                                       "note": "start hold + everyone within 8 m: every ego solves a 4-player game; the two halves of the split batch run on two streams, so the stage totals overlap in time"}}

    if D.rank == 0:
        avg = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}           # ms per launch, HIP events on the handle's stream
        tot = {k: v[0] for k, v in prof.items()}
        dom = max(tot, key=lambda k: tot[k])                                         # the stage with the largest total in THIS run
        dom_ms, dom_n = avg[dom], max(prof[dom][1], 1)
        # players per game over the live egos at the end of the timed region
        live = (st["flags"] & 4) != 0
        dx = st["px"][:, :, None] - st["px"][:, None, :]
        dz = st["pz"][:, :, None] - st["pz"][:, None, :]
        near = (np.sqrt(dx * dx + dz * dz) < 8.0).sum(axis=2)                        # players within 8 m incl. self (HKA:714)
        hist = np.bincount(near[live].ravel(), minlength=A + 1)[1:]
        hist = (hist / max(hist.sum(), 1)).round(4).tolist()
        exec_flop = sum(h * lq_flop(i + 1) for i, h in enumerate(hist)) * A / (4 if A > 2 else 1) + 2000.0 * A
        tick_ms = dt / a.steps * 1e3
        if dom in ("env_run_kernel", "env_b1_kernel"):
            # one launch of the tick kernel advances the envs by a variable number of ticks (<= RUN_CAP): units per launch =
            # env-steps of the timed region / launches of the timed region; algorithmic bytes = 1 056 B x that.
            # A B1 launch (sensing + assembly + the solves of a solve tick; since round 6 also the multi-player games of a spread field, so in a short
            # window with many games it can be the larger stage) passes over the envs of its launch ONCE: state in + state out per env, not per env-step
            if dom == "env_run_kernel":
                algo = ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E * a.steps / dom_n
            else:
                algo = ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E / max(int(sched.get("streams", 1)), 1)
            achieved = algo / 1e9 / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
            traffic, prov, binding = pmc_fields(dom, float(E) * a.steps / dom_n)
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": algo,
                    "traffic_provenance": prov, "binding": binding}
            if dom == "env_b1_kernel":
                run_ms, run_n = avg["env_run_kernel"], max(prof["env_run_kernel"][1], 1)
                run_algo = ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E * a.steps / run_n
                roof["algorithmic_bytes_are"] = "one pass over the launch's envs (1 056 B x envs of the launch): the kernel runs once per solve cadence"
                roof["tick_kernel"] = {"kernel": "env_run_kernel", "launches": prof["env_run_kernel"][1], "avg_launch_ms": run_ms, "algorithmic_bytes_per_launch": run_algo,
                                       "achieved": run_algo / 1e9 / (run_ms * 1e-3) if run_ms > 0 else 0.0,
                                       "frac": (run_algo / 1e9 / (run_ms * 1e-3) / HBM_PEAK_GBS) if run_ms > 0 else 0.0,
                                       "note": "the tick loop's launches of the same window (the larger stage over a long region: see baseline_protocol_ticks_512_3584)"}
            cad = cadence_traffic(float(E) * a.steps / max(prof["env_run_kernel"][1], 1))
            if cad:
                roof["traffic_whole_cadence"] = cad
            if binding:
                vp = valu_port_use(max(prof["env_run_kernel"][1], 1), dt)
                if vp:
                    binding["valu_port_use_over_wall"] = vp
            if sched.get("streams", 1) > 1:
                # the default schedule of a long call runs the batch as two halves on two streams: two tick launches share the GPU, a launch's
                # duration is no longer the kernel's own — the fraction of the roof is the whole job's (every kernel + the gaps), the per-launch
                # figures (they agree with rocprofv3 --kernel-trace --stats of this command) stay beside it
                wj = ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E / 1e9 / (tick_ms * 1e-3)
                roof.update({"per_launch": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": dom_ms,
                                            "note": "half-batch launches of two streams overlap: a launch's duration includes the other half's share of the GPU"},
                             "achieved": wj, "frac": wj / HBM_PEAK_GBS, "overlapped_launches": True,
                             "frac_is": "whole job: algorithmic bytes of the timed region / its wall time"})
        else:
            # the multi-player solver dominates (race start / close racing): price it against the fp64 vector peak with the
            # dense flop count of the games it actually solved in this run (hk_prof_games)
            flop = sum(n * lq_flop(N) for N, n in games.items()) / dom_n
            achieved = flop / 1e12 / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
            roof = {"bound": "fp64_valu", "kernel": "the multi-player solver launch (lqn_round_kernel: pairs + lqn_body<3,4>, or lqn_spread_kernel: see config.schedule)", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "dense_flop_per_launch": flop}
        roof.update({"avg_launch_ms": dom_ms, "launches": prof[dom][1], "chosen_as": "largest total_ms among the stages of the profiled pass",
                     "profiled_pass": {"value": E * D.world * a.steps / dt_prof, "unit": "env-steps/s",
                                       "note": "the same reset / pre-roll / warm-up / ticks once more with hk_prof on: the stage times of this object; `value` is the pass before it, hk_prof off"},
                     "kernel_total_ms": tot, "kernel_avg_ms": avg, "multi_player_games_solved": {str(k): v for k, v in games.items() if v},
                     "whole_job_hbm": {"achieved": ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E / 1e9 / (tick_ms * 1e-3),
                                       "frac": ALGO_BYTES_PER_ENV_STEP * (A / 4.0) * E / 1e9 / (tick_ms * 1e-3) / HBM_PEAK_GBS,
                                       "note": "algorithmic bytes of the timed region / its wall time (all kernels + launch gaps)"},
                     "fp64_valu": {"dense_equivalent_flop_per_env_step_N4": ALGO_FLOP_PER_ENV_STEP,
                                   "executed_flop_per_env_step_at_measured_N": exec_flop,
                                   "achieved_tflops_dense_equivalent": ALGO_FLOP_PER_ENV_STEP * value / D.world / 1e12,
                                   "achieved_tflops_at_measured_N": exec_flop * value / D.world / 1e12,
                                   "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                   "frac_at_measured_N": exec_flop * value / D.world / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                   "note": "SURVEY §8d priced the path at N = 4 players per game; once the field spreads (> 8 m) ~99 % of the games are single-player"}})
        out = {
            "metric": "env-steps/sec (4-agent Oval, batch=65k)" if (A == 4 and E == 65536) else "env-steps/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": D.world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": tick_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d-agent Oval, Fixed-LQNG vs Fixed-LQNG (2v2), %d parallel envs per GPU, seeded start jitter, auto-reset"
                                   % (A, E), "envs_per_gpu": E, "agents": A, "sharding": "envs split contiguously over ranks", "same_device": bool(a.same_device),
                       "ticks": "race ticks %d..%d timed (untimed before: pre-roll 0..%d to the steady state of BASELINE.md §3, then %d warm-up ticks)"
                                % (t_first, t_first + a.steps, a.preroll, a.warmup),
                       "timed_tick_range": [t_first, t_first + a.steps],
                       "players_per_game_hist_N1..": hist, "schedule": sched, "build": hk.build_info(),
                       "finished_episodes_seen": int((results["episode"] >= 0).any(axis=1).sum())},
            "roofline": roof,
        }
        if D.dist:
            # N ranks: every rank's own span of the timed region (value = all ranks' env-steps / the largest), the ranks the process group holds, and
            # the path's one exchange step on its own clock (off the timed region): an all-gather of 32 B per agent from the library's device buffer
            out["multi_gpu"] = {"per_rank_ms": per_rank_ms, "ranks_seen": int(D.dist.get_world_size()), "backend": D.backend,
                                "result_gather": {"ms": gather_ms, "envs_gathered": int(results.shape[0]), "bytes_per_rank": int(E * A * results.dtype.itemsize),
                                                  "note": "torch.distributed all_gather (RCCL on GPUs) straight from the device buffer of hk_episode_result[E][A]; after the timed region"},
                                "note": "no collective and no barrier inside any timed span: a span starts at a common barrier and ends on the rank's own completion"}
        if rep_dt:
            vals = sorted(E * D.world * a.steps / t for t in rep_dt)
            med = vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2])
            out["window_repeats"] = {"n": len(vals), "min": vals[0], "median": med, "max": vals[-1], "unit": "env-steps/s",
                                     "value_over_median": value / med,
                                     "note": "the next %d back-to-back windows of the same %d ticks (race ticks %d..%d), each timed like `value`" % (
                                         len(vals), a.steps, t_first + a.steps, t_first + a.steps * (1 + len(vals)))}
        out.update(secondary)
        if not a.no_cpu_baseline and D.world == 1:          # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(A, seed, a.preroll)
        print(json.dumps(out), flush=True)
    D.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 4 096 ticks per env, steady state = ticks 512-3 584 -> pre-roll 512 (set-up), timed 3 072
    ap.add_argument("--steps", type=int, default=3072)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--preroll", type=int, default=None, help="untimed set-up ticks before the warm-up (default: 512 for the lqng workload = "
                                                                "the start of BASELINE.md's steady-state window, 0 otherwise)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the labelled secondary measurements (protocol window, race start)")
    ap.add_argument("--mcts-iterations", type=int, default=64)
    ap.add_argument("--lq-batch", type=int, default=262144)
    ap.add_argument("--selftest-launcher", action="store_true", help="CPU check of the N-rank plumbing (gloo; no GPU, no libhk compute)")
    ap.add_argument("--same-device", action="store_true", help="every rank on GPU 0 with gloo in place of RCCL (a rehearsal of the N-rank code on a 1-GPU box; "
                                                                "tests/test_two_ranks_one_gpu.py): not a scaling measurement")
    ap.add_argument("--workload", choices=("lqng", "rl", "mcts", "mctsrl", "a8", "lqbatch"), default="lqng",
                    help="lqng: BASELINE.json configs[1] (the headline); rl: 2v2 Oval with the RL low-level actor on device (configs[3] shape); "
                         "mcts: 4-agent Complex track, MCTS-LQNG, 16 384 envs (configs[2]); "
                         "mctsrl: 2v2 OvalDuos, MCTS high level + RL low level on device, 32 768 envs per GPU (configs[3]); "
                         "a8: 8-agent Complex, mixed MCTS-RL vs MCTS-LQNG, 131 072 envs per GPU (configs[4]); "
                         "lqbatch: hk_lq_solve_batch alone, 262 144 games, N = 2 and 4")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.preroll is None:
        a.preroll = STEADY_TICK if a.workload == "lqng" else 0

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # the parent of the ranks: must never touch HIP (a process that has initialised the GPU must not spawn-by-exec)
        raise SystemExit(launch_ranks(a, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (a.gpus, world))
    if a.selftest_launcher:
        raise SystemExit(selftest_launcher(a))

    import torch
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libhk has no CPU fallback")
    D = Dist("gloo" if a.same_device else "nccl")
    if a.same_device:
        D.local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libhk has no CPU fallback")
    import __graft_entry__ as ge
    if D.rank == 0:
        ge.build()
    if D.dist:
        D.dist.barrier()
    import hierarchicalkarting_amd as hk
    if a.workload == "rl":
        return bench_rl(a, D, hk)
    if a.workload in ("mcts", "mctsrl", "a8"):
        return bench_mcts(a, D, hk)
    if a.workload == "lqbatch":
        return bench_lqbatch(a, D, hk)
    return bench_lqng(a, D, hk)


if __name__ == "__main__":
    main()
