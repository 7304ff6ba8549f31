"""hierarchicalkarting_amd — MI355X-native batched kart-racing step + feedback LQ Nash-game solve.

The compute lives in libhk.so (hand-written HIP for gfx950 behind the C ABI of include/hk.h); this package is the
Python host mirror of the reference's operator surface for that path.  Nothing here computes on the CPU."""
from . import _lib  # noqa: F401
from ._lib import HkError  # noqa: F401
from .lq import solve_feedback_lqr, solve_feedback_lqr_batch  # noqa: F401
from .env import RacingEnv  # noqa: F401
from .config import make_config  # noqa: F401
from .policy import Policy  # noqa: F401


def build_info():
    """-> dict: what built the libhk.so this process loaded (hk_build_info: compiler, flags, accepted back-end switches, guard variant per unit)"""
    import json
    return json.loads(_lib.load().hk_build_info().decode())
