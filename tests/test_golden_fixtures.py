"""Host restatements vs golden vectors captured from the reference's importable modules
(tests/golden/make_golden.py)."""
import os

import numpy as np

from deepmimic_mujoco_amd.mocap import euler_from_quaternion_rxyz, quaternion_from_euler_rxyz

G = os.path.join(os.path.dirname(__file__), "golden")


def test_euler_from_quaternion_matches_reference():
    g = np.load(os.path.join(G, "transformations_golden.npz"))
    got = np.array([euler_from_quaternion_rxyz(q) for q in g["quat_xyzw"]])
    # same arithmetic in the same order -> bitwise on this platform; allow 1 ulp-ish slack
    assert np.abs(got - g["euler_rxyz"]).max() <= 1e-15
    assert len(got) > 2000


def test_quaternion_from_euler_matches_reference():
    g = np.load(os.path.join(G, "transformations_golden.npz"))
    e = g["euler_in"]
    got = quaternion_from_euler_rxyz(e[:, 0], e[:, 1], e[:, 2])
    assert np.abs(got - g["quat_out"]).max() <= 1e-15


def test_policy_fixture_reproduces_reference_kat():
    """extracted_policy.py:480-485 known answer with the stored fp32 weights."""
    p = np.load(os.path.join(G, "policy_kat.npz"))
    f = np.tanh(p["kat_obs"] @ p["W0"].astype(np.float64) + p["B0"])
    f = np.tanh(f @ p["W2"].astype(np.float64) + p["B2"])
    a = f @ p["WA"].astype(np.float64) + p["BA"]
    assert np.allclose(a, p["kat_expected"])
    f = np.tanh(p["extra_obs"] @ p["W0"].astype(np.float64) + p["B0"])
    f = np.tanh(f @ p["W2"].astype(np.float64) + p["B2"])
    assert np.allclose(f @ p["WA"].astype(np.float64) + p["BA"], p["extra_act"], atol=1e-6)
