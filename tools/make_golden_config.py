#!/usr/bin/env python3
"""Flatten the REFERENCE's own config classes (legged_robot_config.py, widowGo1_config.py; loaded by
path with the package machinery stubbed, they import nothing from isaacgym) into
tests/golden/widowgo1_config.json, so that the config mirror in wbc_amd/config.py can be checked
field by field on any box."""
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/legged_gym/legged_gym/envs"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


for pkg in ("legged_gym", "legged_gym.envs", "legged_gym.envs.base"):
    sys.modules[pkg] = types.ModuleType(pkg)
    sys.modules[pkg].__path__ = []
load("legged_gym.envs.base.base_config", f"{REF}/base/base_config.py")
load("legged_gym.envs.base.legged_robot_config", f"{REF}/base/legged_robot_config.py")
wg = load("wg_cfg", f"{REF}/widowGo1/widowGo1_config.py")


def flatten(obj, prefix=""):
    out = {}
    for k in dir(obj):
        if k.startswith("_") or k == "init_member_classes":
            continue
        v = getattr(obj, k)
        if isinstance(v, type) or (hasattr(v, "__dict__") and not callable(v) and type(v).__module__ not in ("builtins", "numpy")):
            out.update(flatten(v, prefix + k + "."))
        elif callable(v):
            continue
        else:
            try:
                import numpy as np
                if isinstance(v, np.ndarray):
                    v = v.tolist()
            except ImportError:
                pass
            out[prefix + k] = v
    return out


res = {"WidowGo1RoughCfg": flatten(wg.WidowGo1RoughCfg()), "WidowGo1RoughCfgPPO": flatten(wg.WidowGo1RoughCfgPPO())}
path = os.path.join(HERE, "..", "tests", "golden", "widowgo1_config.json")
json.dump(res, open(path, "w"), indent=0, sort_keys=True, default=float)
print("wrote", path, len(res["WidowGo1RoughCfg"]), len(res["WidowGo1RoughCfgPPO"]))
