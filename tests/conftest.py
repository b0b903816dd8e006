import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    from deepmimic_mujoco_amd.model import load_model
    return load_model()


@pytest.fixture(scope="session")
def clips(model):
    """The four BASELINE clips loaded by the host loader: name -> MocapDM."""
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.mocap import MocapDM
    out = {}
    for name in ["walk", "run", "dance_b", "spinkick"]:
        mc = MocapDM(model=model)
        mc.load_mocap(MotionConfig(name).mocap_path)
        out[name] = mc
    return out


@pytest.fixture(scope="session")
def oracle_clips(clips):
    from oracle.oracle import OracleClip
    return {k: OracleClip(*v.tables()) for k, v in clips.items()}
