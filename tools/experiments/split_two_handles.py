import os, sys, time
sys.path.insert(0, '/root/repo')
import hierarchicalkarting_amd as hk
E=65536
def run(env, n=3072):
    env.synchronize(); t0=time.perf_counter(); env.step(n); env.synchronize(); return E*n/(time.perf_counter()-t0)/1e6
def mk(split):
    if split: os.environ["HK_SPLIT"]="1"
    e=hk.RacingEnv(hk.make_config(E,4,jitter_seed=0x5EED0000))
    os.environ.pop("HK_SPLIT",None)
    e.reset(); e.step(512); e.synchronize(); return e
a=mk(True); print("split alone", run(a))
b=mk(False); print("plain with split handle alive", run(b))
c=mk(True); print("split with two others alive", run(c))
del a, b
d=mk(True); print("split after deleting others", run(d))
