#!/usr/bin/env python3
"""Where the masked lanes of the tick kernel are: lane-participation probes (diagnostic build, -DHK_LANEPROF).

  python tools/lane_profile.py --build       (here: cross-compiles build/libhk_laneprof.so, which travels with gpurun)
  python tools/lane_profile.py [--preroll 512 --ticks 256]      (on the GPU box)

Probe k adds, for every wave that reaches it, 1 to a wave counter and the number of lanes switched on to a lane counter (atomics: the
build is slow, its results are bit-identical).  lanes / (64 x waves) is the lane activity AT that point; waves per wave-tick says how
often a wave pays for the region behind it.  SQ counters (profiles/r03_pmc_summary.json) give the kernel-wide average, 33 of 64."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "build", "libhk_laneprof.so")
NAMES = {23: "tick-loop iteration (any lane of the wave still in the loop)", 0: "... with its env running this iteration", 1: "phase A (phase_begin)",
         2: "phase A: kart within 2.2 m -> 3 ray / capsule tests", 3: "phase B1 entered (every tick)", 4: "B1: solve tick, own-kart staging",
         5: "B1 forward ray: grid cell visited", 6: "B1 forward ray: wall tested", 7: "B1 short rays: wall of the cell's list", 8: "B1 short rays: wall passes the box cull (4 ray tests)",
         9: "B1: single-player assembly", 10: "B1: lq1_solve", 11: "B1: multi-player assembly (per player)", 22: "queue binning block", 12: "phase C (phase_move)",
         13: "C: MoveVehicle", 14: "C: kart-kart narrow phase", 15: "C: wall-contact pass", 16: "C: wall contact, box test of 4 walls", 17: "C: wall contact narrow phase",
         18: "C: Trigger candidate", 19: "C: Trigger box test", 20: "C: Trigger entered (section / lane rules)", 21: "C: planFixed"}
ORDER = [23, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 22, 12, 21, 13, 14, 15, 16, 17, 18, 19, 20]


def build():
    import __graft_entry__ as ge
    ge.build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for u in ge.UNITS:
        o = os.path.join(ROOT, "build", "obj", "laneprof_" + u.replace(".hip", ".o"))
        objs.append(o)
        procs.append(subprocess.Popen([hipcc] + ge.HIPCC_FLAGS + ge._backend_flags(hipcc) + ["-DHK_LANEPROF", "-c", os.path.join(ge.CSRC, u), "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs)
    print("built", LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--preroll", type=int, default=512)
    ap.add_argument("--ticks", type=int, default=256)
    ap.add_argument("--envs", type=int, default=65536)
    a = ap.parse_args()
    if a.build:
        return build()
    if os.environ.get("HK_LIB_PATH") != LIB:
        env = dict(os.environ, HK_LIB_PATH=LIB)          # (a -DHK_STAMPS build dumps its counters with every hk_prof_games)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
        line = [l for l in p.stderr.splitlines() if l.startswith("HK_STAMPS")]
        if p.returncode or not line:
            print(p.stderr[-3000:]); return 1
        v = [int(x) for x in line[-1].split()[1:]][48:]           # game_stats[64:]
        wave_ticks = a.envs * 4 // 64 * a.ticks
        print("wave-ticks of the window: %d (%d envs x %d ticks, 16 envs a wave)" % (wave_ticks, a.envs, a.ticks))
        print("%-72s %14s %12s %10s" % ("probe", "waves reached", "per wave-tick", "lanes on"))
        for k in ORDER:
            lanes, waves = v[2 * k], v[2 * k + 1]
            if waves:
                print("[%2d] %-67s %14d %12.3f %8.1f / 64" % (k, NAMES[k], waves, waves / wave_ticks, lanes / waves))
        return 0
    import hierarchicalkarting_amd as hk
    env = hk.RacingEnv(hk.make_config(a.envs, 4, jitter_seed=0x5EED0000))
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
