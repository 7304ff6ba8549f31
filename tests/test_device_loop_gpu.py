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
print("DEVICE_LOOP_OK")
"""


def test_torch_views_drive_the_env_like_the_host_api():
    # a fresh interpreter: this pytest process has libhk (and with it the system HIP runtime) loaded already, and torch must
    # initialise its own HIP runtime first for the two to share one
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=root, capture_output=True, text=True, timeout=600,
                       env={**os.environ, "PYTHONPATH": root})
    assert r.returncode == 0 and "DEVICE_LOOP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
