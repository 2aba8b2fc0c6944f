#!/usr/bin/env python3
"""Write the packaged widowGo1 asset (deep-whole-body-control_amd/wbc_amd/assets/widowgo1_default.wbcasset): the robot model tables
(tools/extract_model.py's JSON of the URDF) and the shipped WidowGo1RoughCfg resolved into wbc_model / wbc_task_cfg / wbc_curriculum,
plus the DoF and rigid-body names -- what wbc_asset_load hands to a binding that does not use this package's Python
(include/wbc_sim.h). tests/test_host_logic.py fails when the file is stale.   python tools/make_asset.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
from wbc_amd import abi
from wbc_amd.config import WidowGo1RoughCfg

data = abi.asset_bytes(abi.load_default_model(), WidowGo1RoughCfg())
open(abi.DEFAULT_ASSET, "wb").write(data)
print("wrote", abi.DEFAULT_ASSET, len(data), "bytes")
