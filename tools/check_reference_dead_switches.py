#!/usr/bin/env python3
"""Build container only: run the REFERENCE's WidowGo1 (tools/ref_harness) with switches this framework refuses
(abi.UNSUPPORTED_SWITCHES) or ignores (abi.REFERENCE_NO_OPS) and record what the reference itself does with them.
    python tools/check_reference_dead_switches.py > profiles/r04_reference_switches.txt"""
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
import numpy as np
import torch

import harness
from harness import make_reference_env, reference_cfg


def run(tag, mutate, steps=3, act_cols=18):
    cfg = reference_cfg()
    cfg.terrain.tot_rows = 400
    cfg.terrain.transform_y = -400 * cfg.terrain.horizontal_scale / 2
    mutate(cfg)
    try:
        env = make_reference_env(8, seed=3, cfg=cfg, heightfield=None)
        env._backend.ora.set_heightfield(None, 0, 0, 0, 0, 0)
        with torch.inference_mode():
            env.reset()
            env.update_command_curriculum()
            outs = []
            for k in range(steps):
                a = torch.from_numpy((0.3 * np.random.default_rng(k).standard_normal((8, act_cols))).astype(np.float32))
                o = env.step(a)
                outs.append(o[0].numpy().copy())
        print(f"[{tag}] ran {steps} steps; obs shape {outs[-1].shape}")
        return outs
    except Exception as e:      # noqa: BLE001
        tb = traceback.extract_tb(e.__traceback__)
        where = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in tb if "reference" in f.filename][-1:]
        print(f"[{tag}] the reference raises {type(e).__name__}: {str(e).splitlines()[0][:160]}  at {where}")
        return None


base = run("shipped config", lambda c: None)


def aag(c):
    c.control.adaptive_arm_gains = True


run("control.adaptive_arm_gains = True, 18-wide actions", aag)
run("control.adaptive_arm_gains = True, 24-wide actions", aag, act_cols=24)


def noise(c):
    c.noise.add_noise = True


n = run("noise.add_noise = True", noise)
print("   observations identical to the shipped run:", all(np.array_equal(a, b) for a, b in zip(base, n)))


def orient(c):
    c.rewards.scales.orientation = -1.0


run("rewards.scales.orientation = -1 (base-class term)", orient)


def cart(c):
    c.goal_ee.command_mode = "cart"


c2 = run("goal_ee.command_mode = 'cart'", cart)
if c2 is not None:
    print("   observations differ from the shipped run:", not all(np.array_equal(a, b) for a, b in zip(base, c2)))


def nopriv(c):
    c.domain_rand.observe_priv = False


run("domain_rand.observe_priv = False", nopriv)
