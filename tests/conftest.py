import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def robot():
    from wbc_amd import abi
    from wbc_amd.config import WidowGo1RoughCfg
    m = abi.load_default_model()
    cfg = WidowGo1RoughCfg()
    return dict(model=m, wmodel=abi.fill_model(m), cfg=cfg, tcfg=abi.fill_task_cfg(cfg, m))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()
