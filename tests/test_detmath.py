"""include/hk_detmath.h (the pinned transcendental functions) vs an 80-digit mpmath reference: <= 2 ulp."""
import math
import numpy as np
import mpmath as mp
from oracle_lib import lib

mp.mp.dps = 60


def ulp_err(got, exact):
    exact_f = float(exact)
    if exact_f == 0.0:
        return abs(got) / 5e-324
    u = math.ulp(exact_f)
    return abs(mp.mpf(got) - exact) / u


def test_sin_cos():
    L = lib(); rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-8, 14, 4000), rng.uniform(-1e3, 1e3, 500), [0.0, 1e-9, -1e-9, math.pi / 2, math.pi, 2 * math.pi]])
    worst = 0
    for x in xs:
        x = float(x)
        es, ec = mp.sin(mp.mpf(x)), mp.cos(mp.mpf(x))
        # near a zero of the function an absolute bound is the meaningful one
        for got, ex in ((L.hko_sin(x), es), (L.hko_cos(x), ec)):
            if abs(ex) > 1e-3:
                worst = max(worst, float(ulp_err(got, ex)))
            else:
                assert abs(mp.mpf(got) - ex) < 1e-18 * max(1.0, abs(x))
    assert worst <= 2.0, worst


def test_atan2():
    L = lib(); rng = np.random.default_rng(2)
    worst = 0
    pts = [(float(a), float(b)) for a, b in rng.uniform(-60, 60, (5000, 2))]
    pts += [(0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (1e-9, 1.0), (1.0, 1e-9), (3.0, -4.0)]
    for y, x in pts:
        ex = mp.atan2(mp.mpf(y), mp.mpf(x))
        got = L.hko_atan2(y, x)
        if ex == 0:
            assert got == 0.0
        else:
            worst = max(worst, float(ulp_err(got, ex)))
    assert worst <= 2.0, worst
    assert L.hko_atan2(0.0, 0.0) == 0.0
    assert L.hko_atan2(0.0, -1.0) == math.pi
    assert L.hko_atan2(-1.0, 0.0) == -math.pi / 2


def test_exp():
    L = lib(); rng = np.random.default_rng(3)
    worst = 0
    for x in np.concatenate([rng.uniform(-5, 1, 3000), rng.uniform(-700, 700, 300), [0.0]]):
        x = float(x)
        worst = max(worst, float(ulp_err(L.hko_exp(x), mp.exp(mp.mpf(x)))))
    assert worst <= 2.0, worst


def test_log():
    L = lib(); rng = np.random.default_rng(5)
    worst = 0
    for x in np.concatenate([rng.uniform(0.01, 4, 3000), 10.0 ** rng.uniform(-20, 20, 300), [1.0, 0.8125, 2.0]]):
        x = float(x)
        ex = mp.log(mp.mpf(x))
        got = L.hko_log(x)
        if abs(ex) > 1e-3:
            worst = max(worst, float(ulp_err(got, ex)))
        else:
            assert abs(mp.mpf(got) - ex) < 1e-18
    assert worst <= 2.0, worst


def test_matches_libm_within_float_rounding():
    """Mathf.* = (float)libm(double): the pinned functions and glibc agree after rounding to float except on a
    vanishing fraction of inputs (the reference's own libm is unknowable; see hk_detmath.h)."""
    L = lib(); rng = np.random.default_rng(4)
    xs = rng.uniform(-7, 7, 20000)
    diff = sum(np.float32(L.hko_sin(float(x))) != np.float32(math.sin(float(x))) for x in xs)
    assert diff <= 2


def _float_api(L):
    import ctypes as C
    for n, args in (("hko_sinf", 1), ("hko_cosf", 1), ("hko_expf", 1), ("hko_atan2f", 2)):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_float] * args
    fp = C.POINTER(C.c_float)
    for n in ("hko_sincosf", "hko_sincosf_near0"):
        getattr(L, n).restype = None
        getattr(L, n).argtypes = [C.c_float, fp, fp]
    return L


def test_float_entry_points_are_the_correctly_rounded_float_almost_always():
    """Mathf.Sin / Cos / Atan2 / Exp of the per-tick physics: evaluated in double to <= 2^-44 relative and rounded once.  Against mpmath: never
    more than one float ulp away, and equal to the correctly rounded float — what (float)Math.Sin((double)x) returns under any libm — on all
    but a ~1e-6 fraction of the arguments (where the true value sits within 2^-44 of a rounding boundary)."""
    L = _float_api(lib()); rng = np.random.default_rng(12)

    def check(fn, exact, args):
        wrong = 0
        for a in args:
            a = tuple(float(np.float32(v)) for v in (a if isinstance(a, tuple) else (a,)))
            got = np.float32(fn(*a))
            ex = exact(*[mp.mpf(v) for v in a])
            want = np.float32(float(ex))               # (double rounding: mpmath -> double -> float; a 2^-29 fraction of the cases could differ)
            if got != want:
                wrong += 1
                assert abs(float(got) - float(ex)) <= 1.0001 * float(np.spacing(np.abs(want))), (a, got, want)
        return wrong
    xs = np.concatenate([rng.uniform(-7, 7, 6000), rng.uniform(-0.6, 0.6, 3000), rng.uniform(-400, 400, 1000), [0.0, 1e-6, -1e-6, 0.5235988, 6.2831855]])
    assert check(L.hko_sinf, mp.sin, xs) <= 1 and check(L.hko_cosf, mp.cos, xs) <= 1
    assert check(L.hko_expf, mp.exp, np.concatenate([rng.uniform(-3, 1, 6000), rng.uniform(-80, 80, 500), [0.0]])) <= 1
    pts = [(a, b) for a, b in rng.uniform(-60, 60, (6000, 2))] + [(0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (3.0, -4.0)]
    assert check(L.hko_atan2f, mp.atan2, pts) <= 1
    assert L.hko_atan2f(0.0, 0.0) == 0.0 and L.hko_expf(0.0) == 1.0 and L.hko_sinf(0.0) == 0.0 and L.hko_cosf(0.0) == 1.0
    assert L.hko_expf(-1000.0) == 0.0


def test_float_sincos_pair_and_small_angle_path_are_bit_identical():
    """the kernels call hk_sincosf / hk_sincosf_near0 where the oracle calls hk_sinf and hk_cosf: same bits"""
    import ctypes as C
    L = _float_api(lib()); r = np.random.default_rng(13)
    xs = np.concatenate([r.uniform(-7, 7, 20000), r.uniform(-1e3, 1e3, 3000), r.uniform(-0.79, 0.79, 20000),
                         [0.0, -0.0, 0.78, -0.78, np.nextafter(np.float32(0.78), np.float32(0)), np.nextafter(np.float32(0.78), np.float32(1)), 0.7853982]]).astype(np.float32)
    s, c, s0, c0 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for x in xs:
        L.hko_sincosf(float(x), C.byref(s), C.byref(c))
        L.hko_sincosf_near0(float(x), C.byref(s0), C.byref(c0))
        a, b = np.float32(L.hko_sinf(float(x))), np.float32(L.hko_cosf(float(x)))
        for u, v in ((s.value, a), (c.value, b), (s0.value, a), (c0.value, b)):
            assert np.float32(u).view(np.uint32) == v.view(np.uint32), x


def test_expf_fast_within_2ulp_and_saturates():
    """the fp32 exp of the RL actor's Swish: <= 2 ulp on its clamped domain, monotone at the clamps"""
    import ctypes as C
    L = lib()
    L.hko_expf_fast.restype = C.c_float
    L.hko_expf_fast.argtypes = [C.c_float]
    r = np.random.default_rng(0)
    xs = np.concatenate([r.uniform(-87, 88, 20000), r.uniform(-2, 2, 20000), [0.0, -0.0, 1.0, -1.0, 88.0, -87.0]]).astype(np.float32)
    got = np.array([L.hko_expf_fast(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - ref) / ulp).max() <= 2.0
    assert L.hko_expf_fast(1000.0) == L.hko_expf_fast(88.0) and L.hko_expf_fast(-1000.0) == L.hko_expf_fast(-87.0)
    assert L.hko_expf_fast(0.0) == 1.0


def test_sincos_pair_is_bit_identical_to_sin_and_cos():
    """the HIP kernels use hk_sincos where they need both values; the oracle calls hk_sin / hk_cos: they must agree exactly"""
    import ctypes as C
    L = lib()
    L.hko_sincos.restype = None
    L.hko_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    r = np.random.default_rng(11)
    xs = np.concatenate([r.uniform(-7, 7, 20000), r.uniform(-1e4, 1e4, 5000), r.uniform(-1e-3, 1e-3, 2000),
                         [0.0, -0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 0.78539816339744828]])
    xs = np.concatenate([xs, xs.astype(np.float32).astype(np.float64)])
    L.hko_sincos_near0.restype = None
    L.hko_sincos_near0.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    xs = np.concatenate([xs, r.uniform(-0.79, 0.79, 20000), [0.78, -0.78, np.nextafter(0.78, 0), np.nextafter(0.78, 1), 0.7853981633974483]])
    s, c = C.c_double(), C.c_double()
    s0, c0 = C.c_double(), C.c_double()
    for x in xs:
        L.hko_sincos(float(x), C.byref(s), C.byref(c))
        assert s.value == L.hko_sin(float(x)) and c.value == L.hko_cos(float(x)), x
        L.hko_sincos_near0(float(x), C.byref(s0), C.byref(c0))            # the reduction-free path for small angles: same bits
        assert s0.value == s.value and c0.value == c.value and np.signbit(s0.value) == np.signbit(s.value), x
        assert (s.value == 0.0) == (L.hko_sin(float(x)) == 0.0)          # also the sign of zero
        assert np.signbit(s.value) == np.signbit(L.hko_sin(float(x))) and np.signbit(c.value) == np.signbit(L.hko_cos(float(x)))


def test_float_entry_points_special_cases():
    """signed zero, huge and non-finite arguments of the float API: the sign of atan2 follows y's SIGN BIT (Math.Atan2(-0, x < 0) = -pi),
    arguments beyond the fast reduction's range go through the double kernels, a non-finite argument gives a non-finite sine / cosine and never an out-of-range table index"""
    L = _float_api(lib())
    pi = float(np.float32(math.pi))
    assert np.float32(L.hko_atan2f(-0.0, -1.0)) == np.float32(-math.pi) and np.float32(L.hko_atan2f(0.0, -1.0)) == np.float32(math.pi)
    r = np.float32(L.hko_atan2f(-0.0, 1.0))
    assert r == 0.0 and np.signbit(r) and not np.signbit(np.float32(L.hko_atan2f(0.0, 1.0)))
    assert abs(L.hko_atan2f(-1e-30, -1.0) + pi) < 1e-6
    for x in (1.5e5, -3.3e6, 1.0e9, 16777216.0):
        xf = float(np.float32(x))
        assert abs(L.hko_sinf(xf) - float(mp.sin(mp.mpf(xf)))) < 2e-7 and abs(L.hko_cosf(xf) - float(mp.cos(mp.mpf(xf)))) < 2e-7, x
    for bad in (float("nan"), float("inf"), -float("inf")):
        assert not math.isfinite(L.hko_sinf(bad)) and not math.isfinite(L.hko_cosf(bad))      # (NaN or inf: the kart's f_finite check catches either)
        assert math.isnan(L.hko_atan2f(bad, 1.0)) or abs(L.hko_atan2f(bad, 1.0)) <= 1.6
        v = L.hko_atan2f(bad, bad)
        assert math.isnan(v) or abs(v) <= 3.2
