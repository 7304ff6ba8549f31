"""Why did the two-halves secondary of bench.py read 1 270 M while the main handle was alive (and 1 467 M once it was closed)?
usage: python tools/experiments/split_two_handles.py [torch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "torch" in sys.argv:
    import torch
    torch.zeros(4, device="cuda"); torch.cuda.synchronize()
import hierarchicalkarting_amd as hk
E = 65536
def run(env, n=3072):
    env.synchronize(); t0 = time.perf_counter(); env.step(n); env.synchronize(); return E * n / (time.perf_counter() - t0) / 1e6
def mk(split):
    if split: os.environ["HK_SPLIT"] = "1"
    e = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
    os.environ.pop("HK_SPLIT", None)
    e.reset(); e.step(512); e.synchronize(); return e
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " torch:", "torch" in sys.argv)
a = mk(False); print("plain handle (its race start used both of its streams)  %7.1f M" % run(a))
b = mk(True); print("split handle, plain handle alive                        %7.1f M" % run(b))
a.close(); print("split handle, plain handle closed                       %7.1f M" % run(b))
