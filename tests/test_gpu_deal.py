"""The step kernel deals the envs of an XCD's range to its workgroups per launch (balanced: the robots expected to be in contact go to
different SIMDs; csrc/wbc_step_kernel.hip, header comment of wbc_step_kernel). Which workgroup steps an env must not show anywhere: a
sim that deals and one that does not (WBC_NO_DEAL=1 at creation) agree BIT FOR BIT on every tensor after 60 steps with resets, at every
size the dealing is on for -- which also proves the deal a bijection launch after launch (an env taken twice or not at all would differ)."""
import os

import numpy as np
import pytest
import torch

import helpers
from wbc_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1024, 2048, 2560, 4096])       # (the deal is on from 2048 envs; 1024: both sims undealt, the test's own baseline)
def test_dealt_and_undealt_sims_agree_bit_for_bit(robot, n):
    params = helpers.random_env_params(n, 3)
    sims = []
    for off in ("0", "1"):
        os.environ["WBC_NO_DEAL"] = off
        try:
            g = helpers.make_gpu(robot, n, params, seed=5)
        finally:
            os.environ.pop("WBC_NO_DEAL", None)
        g.reset_all()
        sims.append(g)
    gen = torch.Generator(device="cuda").manual_seed(11)
    resets = 0
    for i in range(60):
        a = torch.randn(n, 18, device="cuda", generator=gen) * 0.6
        for g in sims: g.step(a)
        resets += int(sims[0].tensor("RESET_BUF").sum().item())
        if i % 10 == 9 or i < 3:
            torch.cuda.synchronize()
            for name in abi.TENSOR_IDS:
                x, y = sims[0].tensor(name), sims[1].tensor(name)
                assert torch.equal(x, y), (n, i, name, int((x != y).sum().item()))
    assert resets > n // 4        # contacts, terminations and resets happened: the hints were not all zero
