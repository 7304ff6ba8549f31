"""Device-resident RL loop: torch tensors aliasing libhk's buffers (observations, rewards, actions) drive the env without
any host copy, and give the same trajectory as the host-copy API."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu


SCRIPT = r"""
import torch
torch.cuda.init()                      # torch's HIP runtime first: libhk then binds to the same libamdhip64 (see RacingEnv.torch_views)
import numpy as np
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
kw = dict(num_envs=64, num_agents=2, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_RL], rewards=1, jitter_seed=4)
a = hk.RacingEnv(hk.make_config(**kw)); b = hk.RacingEnv(hk.make_config(**kw))
a.reset(); b.reset()
v = a.torch_views()
assert v["obs"].is_cuda and v["obs"].shape == (64, 2, a.obs_dim)
gen = torch.Generator(device="cpu"); gen.manual_seed(0)
tot_a = np.zeros((64, 2)); tot_b = np.zeros((64, 2))
for k in range(60):
    steer = (torch.rand((64, 2), generator=gen) * 2 - 1)
    branch = torch.randint(0, 3, (64, 2), generator=gen, dtype=torch.int32)
    # device path: observe -> (a policy would read v["obs"] here) -> actions written in place -> step -> rewards on device
    a.observe(); a.synchronize()
    obs_a = v["obs"].clone()
    v["act_steer"].copy_(steer.cuda()); v["act_branch"].copy_(branch.cuda())
    torch.cuda.synchronize()
    a.step(2)
    a.rewards_device(); a.synchronize()
    tot_a += v["reward"].cpu().numpy()
    # host path
    obs_b = b.observations()
    b.set_actions(steer.numpy(), branch.numpy())
    b.step(2)
    tot_b += b.rewards()[0]
    assert np.array_equal(obs_a.cpu().numpy(), obs_b), k
sa, sb = a.agent_state(), b.agent_state()
for name in sa.dtype.names:
    assert np.array_equal(sa[name], sb[name]), name
assert np.array_equal(tot_a, tot_b) and (tot_a != 0).all()
# hk_lq_solve_batch_device: the LQ solve on tensors that already live on the GPU (no host staging), same bits as the host entry point
import ctypes as C
from oracle import lq_numpy as LQ
rng = np.random.default_rng(3)
games = [LQ.random_game(rng, 3) for _ in range(300)]
args = [np.ascontiguousarray(np.array([g[k] for g in games]), np.float64) for k in range(6)]
u_host = hk.solve_feedback_lqr_batch(*args, 3)
dev = [torch.from_numpy(x).cuda() for x in args]
u_dev = torch.zeros((300, 2), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
rc = a.L.hk_lq_solve_batch_device(a.h, 300, 3, dp(dev[0]), dp(dev[1]), dp(dev[2]), dp(dev[3]), dp(dev[4]), dp(dev[5]), 3, dp(u_dev), None)
assert rc == 0, rc
a.synchronize()
assert np.array_equal(u_dev.cpu().numpy().view(np.uint64), u_host.view(np.uint64))
print("DEVICE_LOOP_OK")
"""


def test_torch_views_drive_the_env_like_the_host_api():
    # a fresh interpreter: this pytest process has libhk (and with it the system HIP runtime) loaded already, and torch must
    # initialise its own HIP runtime first for the two to share one
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=root, capture_output=True, text=True, timeout=600,
                       env={**os.environ, "PYTHONPATH": root})
    assert r.returncode == 0 and "DEVICE_LOOP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
