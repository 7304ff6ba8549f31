"""A plain-C host (examples/hk_host.c) linked against libhk.so — no Python, no ctypes in the product path: the records and
observations it dumps equal the CPU oracle's bit for bit.  This is the shape of the C# P/Invoke host of INTEGRATION.md."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.env import AGENT_DT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_host(tmp_path):
    exe = str(tmp_path / "hk_host")
    libdir = os.path.join(ROOT, "hierarchicalkarting_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "examples", "hk_host.c"),
                           "-L", libdir, "-lhk", "-Wl,-rpath," + libdir])
    return exe


def test_c_host_compiles_and_links(tmp_path):
    """(no GPU needed) the header is valid C and every symbol the driver uses resolves against libhk.so"""
    import __graft_entry__ as ge
    ge.build()
    exe = _build_host(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_host_matches_the_oracle(tmp_path):
    import hierarchicalkarting_amd as hk
    import __graft_entry__ as ge
    ge.build()
    exe = _build_host(tmp_path)
    b = hk.make_config(7, 4, jitter_seed=0x5EED0000)
    blob = tmp_path / "cfg.bin"
    with open(blob, "wb") as f:
        f.write(bytes(b.cfg))
        f.write(bytes(b.sections))
        f.write(bytes(b.walls))
    out = tmp_path / "out.bin"
    ticks, calls = 40, 6
    r = subprocess.run([exe, str(blob), str(ticks), str(calls), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = O.OracleEnv(b)
    o.reset()
    raw = np.fromfile(out, np.uint8)
    E, A = 7, 4
    obs_dim = 9 + 5 * b.cfg.section_horizon + 8 + 12 * (A - 1)
    rec = E * A * AGENT_DT.itemsize + E * A * obs_dim * 4
    assert raw.size == calls * rec
    for c in range(calls):
        o.step(ticks)
        chunk = raw[c * rec:(c + 1) * rec]
        st = chunk[:E * A * AGENT_DT.itemsize].view(AGENT_DT).reshape(E, A)
        ob = chunk[E * A * AGENT_DT.itemsize:].view(np.float32).reshape(E, A, obs_dim)
        os_ = o.agent_state()
        for name in AGENT_DT.names:
            assert np.array_equal(st[name], os_[name]), (c, name)
        assert np.array_equal(ob.view(np.uint32), o.observations().view(np.uint32)), c
