"""hk_step(1) back to back from the steady state, for rocprofv3 --kernel-trace --stats: which kernels a one-tick call is made of and how long each takes.
usage (GPU box):  rocprofv3 --kernel-trace --stats -d gpurun_out/step1 -o t --output-format csv -- python3 tools/experiments/step1_trace.py [n_ticks_per_call]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
env = hk.RacingEnv(hk.make_config(65536, 4, jitter_seed=0x5EED0000))
env.reset(); env.step(512); env.synchronize()
t0 = time.perf_counter()
for _ in range(256):
    env.step(n)
env.synchronize()
print("hk_step(%d): %.1f us per call" % (n, (time.perf_counter() - t0) / 256 * 1e6))
