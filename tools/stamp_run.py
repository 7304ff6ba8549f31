#!/usr/bin/env python3
"""diagnostic: average in-kernel cycles per phase of env_assemble_kernel (HK_LQ_DEBUG=128)"""
import ctypes as C, os, sys
os.environ["HK_LQ_DEBUG"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hierarchicalkarting_amd as hk
env = hk.RacingEnv(hk.make_config(65536, 4, jitter_seed=0x5EED0000))
env.reset(); env.step(int(sys.argv[1]) if len(sys.argv) > 1 else 800); env.synchronize()
out = (C.c_uint64 * 16)()
env.L.hk_debug_cycles(env.h, out)
n = max(out[5], 1)
print("waves stamped", out[5])
for i, nm in enumerate(["load+derive", "rays", "sync+players", "assemble_player", "lq1_solve"]):
    print("%-16s %10.0f cycles/wave" % (nm, out[i] / n))
