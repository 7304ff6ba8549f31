#!/usr/bin/env python3
"""Phase profile of the fused tick kernel from in-kernel cycle stamps (diagnostic build, -DHK_STAMPS).

  python tools/stamp_profile.py --build              (here: cross-compiles build/libhk_stamps.so, which travels with gpurun)
  python tools/stamp_profile.py [--preroll 512 --ticks 512 --envs 65536 --agents 4]     (on the GPU box)

Each lane accumulates the cycles between consecutive stamps; at the end of a launch the wave's maximum per counter is added
to a device array.  The shares below are therefore wave time (two waves share a SIMD, so they include the partner's issue
slots), summed over the waves that entered the tick loop.  A lane that skips stamps (an env that is not on a solve tick while
its wave neighbours are) books the whole skipped stretch on the next stamp it executes, and the wave's maximum picks that up: the
counters behind a divergent stretch ([2] after [14] / [15], [6] after [3] .. [5]) are upper bounds.  (Round 2: [2] looked like
12 % in four 2 m rays; making those rays division-free changed neither the stamp nor the kernel's time.)"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "build", "libhk_stamps.so")
NAMES = ["prologue (tables -> LDS, state load)", "phase A: episode controller, kart-vs-kart rays", "own-kart staging + 5 wall rays",
         "players within 8 m", "single-player assembly (heading heuristic)", "lq1_solve", "queue binning", "multi-player assembly",
         "actions, planFixed, ArcadeKart, integrate", "kart-kart contacts", "kart-wall contacts", "Triggers, section / lane rules",
         "telemetry, env words", "wait for the wave's other groups", "(of [2]) own-kart staging: atan2, max speed, Trigger distance", "(of [2]) forward wall ray", "(of [2]) four short rays", "(of [2]) KartS -> LDS", "B1 kernel: queue binning, stores (waited for)", "B1 kernel: table staging, record loads, sincos",
         "tick kernel head: lane group + env words", "tick kernel head: table staging", "tick kernel tail: record stores (waited for)", "B1 kernel head: lane group + env words",
         "tick kernel: a wave's whole life in a launch (not in the %)", "B1 kernel: a wave's whole life in a launch (not in the %)",
         "(of [11]) finite checks, capsule core, candidate cell", "(of [11]) Trigger overlap tests", "(spare)", "(spare)"]
WHOLE = (24, 25)
NST = len(NAMES)


def build():
    import __graft_entry__ as ge
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for u in ge.UNITS:
        o = os.path.join(ROOT, "build", "obj", "stamps_" + u.replace(".hip", ".o"))
        objs.append(o)
        procs.append(subprocess.Popen([hipcc] + ge.HIPCC_FLAGS + ge._backend_flags(hipcc) + ["-DHK_STAMPS"] + os.environ.get("HK_STAMPS_EXTRA", "").split() + ["-c", os.path.join(ge.CSRC, u), "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    objs.append(os.path.join(ROOT, "build", "obj", "hk_build_info.o"))          # (hk_build_info(): the product's record; ge.build() above... made it)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs)
    print("built", LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--preroll", type=int, default=512)
    ap.add_argument("--ticks", type=int, default=512)
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--agents", type=int, default=4)
    a = ap.parse_args()
    if a.build:
        return build()
    if os.environ.get("HK_LIB_PATH") != LIB:            # re-run with the diagnostic library and capture its dump
        env = dict(os.environ, HK_LIB_PATH=LIB)          # (a -DHK_STAMPS build dumps its counters with every hk_prof_games)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
        line = [l for l in p.stderr.splitlines() if l.startswith("HK_STAMPS")]
        if p.returncode or not line:
            print(p.stderr[-3000:]); return 1
        v = [int(x) for x in line[-1].split()[1:]]
        tot = sum(v[k] for k in range(NST) if k not in WHOLE) or 1
        print("waves of the tick kernel that entered the loop (summed over launches): %d; waves of the B1 kernel with work: %d" % (v[NST], v[NST + 1]))
        B1 = (2, 3, 4, 5, 6, 7, 14, 15, 16, 17, 18, 19, 23, 25)        # (with the fission these phases run in the B1 kernel: per wave-launch of THAT kernel)
        for k in range(NST):
            waves = v[NST + 1] if (k in B1 and v[NST + 1]) else v[NST]
            print("  [%2d] %-50s %6.2f %%   %8.1f kcycles / wave-launch" % (k, NAMES[k], 100.0 * v[k] / tot, v[k] / max(waves, 1) / 1e3))
        return 0
    import hierarchicalkarting_amd as hk
    env = hk.RacingEnv(hk.make_config(a.envs, a.agents, jitter_seed=0x5EED0000))
    env.reset()
    if a.preroll:
        env.step(a.preroll)
    env.synchronize()
    env.prof_reset()
    env.step(a.ticks)
    env.synchronize()
    env.prof_games()


if __name__ == "__main__":
    sys.exit(main())
