"""HK_SPLIT=1: the batch as two halves on two streams (hk_api.hip issue_rounds_split) must change nothing but the speed.
The switch is read once per process, so the comparison runs in a child process."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
b = hk.make_config(8192 + 64, 4, jitter_seed=0x5EED0000)          # (not a multiple of 128: the halves are unequal)
g = hk.RacingEnv(b); o = O.OracleEnv(b)
g.reset(); o.reset()
t = 0
for n in (70, 130, 20, 7, 100):                                    # start hold, race start in packs, short and long calls
    g.step(n); o.step(n); t += n
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name)
print("split ok", t)
"""


@pytest.mark.gpu
def test_split_batch_matches_the_oracle():
    env = dict(os.environ, HK_SPLIT="1")
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "split ok 327" in r.stdout
