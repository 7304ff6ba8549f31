"""csrc/hk_swish.h (v_med3_f32 clamp) == include/hk_detmath.h hk_swishf (plain compares) for every fp32 input, on the GPU;
and the oracle's C build of hk_swishf against a float64 evaluation."""
import os, subprocess, ctypes as C
import numpy as np
import pytest
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swish_contract_on_cpu():
    L = O.lib()
    L.hko_swishf.restype = C.c_float
    L.hko_swishf.argtypes = [C.c_float]
    xs = np.concatenate([np.linspace(-100, 100, 4001), [-88.0, -87.999, -88.001, 87.0, 0.0, 1e-30, -1e-30]]).astype(np.float32)
    got = np.array([L.hko_swishf(float(x)) for x in xs], np.float64)
    xd = xs.astype(np.float64)
    want = xd / (1.0 + np.exp(np.clip(-xd, -87.0, 88.0)))
    assert np.all(np.abs(got - want) <= 4e-7 * np.abs(want) + 1e-37)
    assert L.hko_swishf(0.0) == 0.0 and L.hko_swishf(100.0) == 100.0
    assert abs(L.hko_swishf(-95.0)) < 1e-35                         # exp argument held at 88


@pytest.mark.gpu
def test_device_swish_equals_the_contract_for_every_float(tmp_path):
    exe = str(tmp_path / "swish_check")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                           os.path.join(ROOT, "tests", "swish_device_check.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("mismatches 0 "), r.stdout
