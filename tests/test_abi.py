"""The C-ABI library loads on a CPU-only box and exports every symbol include/hk.h declares; the ctypes struct
layouts equal the C ones; without a GPU the product refuses to compute (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from hierarchicalkarting_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(built):
    hdr = open(os.path.join(ROOT, "include", "hk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(hk_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    L = built.load()
    for n in sorted(names):
        assert hasattr(L, n), "libhk.so does not export %s" % n
        assert n in built.SYMBOLS, "ctypes binding missing for %s" % n
    assert set(built.SYMBOLS) == names


def test_struct_layouts_match_c(built, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(hk_config),sizeof(hk_agent_state),sizeof(hk_env_state),sizeof(hk_episode_result),sizeof(hk_lq_debug),'
                   'sizeof(hk_section),sizeof(hk_kart_stats),offsetof(hk_agent_state,plan_lane),offsetof(hk_config,sections),'
                   'offsetof(hk_config,stats));return 0;}\n' % os.path.join(ROOT, "include", "hk.h"))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(built.Config), C.sizeof(built.AgentState), C.sizeof(built.EnvState), C.sizeof(built.EpisodeResult),
            C.sizeof(built.LqDebug), C.sizeof(built.Section), C.sizeof(built.KartStats), built.AgentState.plan_lane.offset,
            built.Config.sections.offset, built.Config.stats.offset]
    assert got == want


def test_no_cpu_fallback(built):
    """On a box without a HIP device every compute entry point fails loudly with HK_ERR_NO_DEVICE."""
    L = built.load()
    h = C.c_void_p()
    rc = L.hk_create(None, C.byref(h))
    if rc == 0:
        L.hk_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == built.HK_ERR_NO_DEVICE
    assert b"no HIP device" in L.hk_last_error(None)
    import numpy as np
    import hierarchicalkarting_amd as hk
    with pytest.raises(built.HkError) as e:
        hk.solve_feedback_lqr_batch(np.zeros((1, 1, 4, 4)), np.zeros((1, 1, 4, 2)), np.eye(4)[None, None], np.zeros((1, 1, 4)),
                                    np.eye(2)[None, None], np.zeros((1, 4)))
    assert e.value.code == built.HK_ERR_NO_DEVICE
    with pytest.raises(built.HkError):
        hk.RacingEnv(num_envs=2, num_agents=2)


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under the package or the C ABI may include / import / link it"""
    for d, _, files in os.walk(os.path.join(ROOT, "hierarchicalkarting_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(d, f), errors="replace").read()
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', txt), os.path.join(d, f)
                assert not re.search(r'^\s*(from|import)\s+[\w.]*oracle', txt, flags=re.M), os.path.join(d, f)
                assert "hko_" not in txt, os.path.join(d, f)
    out = subprocess.run(["ldd", os.path.join(ROOT, "hierarchicalkarting_amd", "libhk.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
