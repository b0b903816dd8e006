"""The retargeting tool (src/retarget.py:42-194) against the reference's own outputs.

tests/golden/retarget_golden.npz holds the frames of src/mujoco/motions/unitree_g1_{run,walk,getup_facedown}.txt — the
files the reference's tool wrote from humanoid3d_{run,walk,getup_facedown}.txt — plus Euler helper vectors produced by
src/transformations.py for the two axis orders the tool uses (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from deepmimic_mujoco_amd import retarget as R
from deepmimic_mujoco_amd.mocap import MocapDM

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "retarget_golden.npz"))


def test_euler_helpers_match_reference_vectors():
    for e, m, y in zip(G["euler_rxyz_in"], G["matrix_rxyz"], G["euler_ryxz_out"]):
        M = R.euler_matrix_rxyz(*e)
        assert np.abs(M - m).max() < 1e-14
        assert np.abs(np.array(R.euler_from_matrix_ryxz(m)) - y).max() < 1e-12
    # and the matrix is what its name says: Rx(a) Ry(b) Rz(c)
    a, b, c = 0.3, -0.7, 1.1
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    assert np.abs(R.euler_matrix_rxyz(a, b, c) - rx @ ry @ rz).max() < 1e-15
    y, x, z = R.euler_from_matrix_ryxz(rx @ ry @ rz)
    ry2 = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    rx2 = np.array([[1, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    rz2 = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1]])
    assert np.abs(ry2 @ rx2 @ rz2 - rx @ ry @ rz).max() < 1e-14


def test_g1_joint_table_matches_the_reference_clip_header():
    g1 = R.g1_joint_table()
    assert g1["names"] == [str(n) for n in G["joint_names"]] and g1["nq"] == 44 and len(g1["names"]) == 38
    assert g1["qpos_addr"]["floating_base_joint"] == (0, 7) and g1["qpos_addr"]["left_hip_pitch_joint"] == 7
    assert g1["qpos_addr"]["right_six_joint"] == 43
    assert g1["range"]["left_hip_pitch_joint"] == (-2.35, 3.05)      # deepmimic_unitree_g1.xml:107


@pytest.mark.parametrize("motion,shoulders", [("run", False), ("walk", False), ("getup_facedown", True)])
def test_retargeted_frames_equal_the_reference_files(motion, shoulders):
    """All 45 columns of every frame of the reference's three retargeted clips.  unitree_g1_run / _walk were written before
    the tool's shoulder block existed (their shoulder columns are the humanoid's x / y / z angles verbatim): the naive
    mapping reproduces them.  unitree_g1_getup_facedown was written by the tool as it is now — shoulder block, its
    getup-only pitch hack and its two indexing quirks included — and is reproduced with the block on: 183 frames."""
    dt, frames, loop, g1 = R.retarget_frames(motion, shoulder_euler_conversion=shoulders)
    want = G[motion + "_frames"]
    got = np.array(frames)
    assert got.shape == want.shape
    err = np.abs(got - want)
    print(motion, "frames", got.shape, "max |diff|", err.max())
    assert err.max() < 1e-9
    assert str(loop) == str(G[motion + "_loop"])


@pytest.mark.parametrize("motion", ["run", "walk"])
def test_shoulder_block_invariants(motion):
    """The current tool's shoulder re-ordering (retarget.py:75-167): only the six shoulder columns change; every angle
    respects its joint range and the 15 rad/s continuity window; where nothing had to be clipped the G1 triple is the same
    rotation as the (x, y', z'') triple the block read: Ry(pitch-slot) Rx(roll-slot) Rz(yaw) == Rx(hr) Ry(hp) Rz(hy)."""
    dt, naive, _, g1 = R.retarget_frames(motion, shoulder_euler_conversion=False)
    _, full, _, _ = R.retarget_frames(motion)
    naive, full = np.array(naive)[:, 1:], np.array(full)[:, 1:]
    sh = [g1["qpos_addr"]["%s_shoulder_%s_joint" % (s, p)] for s in ("left", "right") for p in ("roll", "pitch", "yaw")]
    other = [c for c in range(44) if c not in sh]
    assert np.array_equal(naive[:, other], full[:, other]) and not np.allclose(naive[:, sh], full[:, sh])
    from deepmimic_mujoco_amd.model import load_model
    hm = load_model()
    hadr = {n: int(hm.jnt_qposadr[i]) for i, n in enumerate(hm.jnt_names)}
    exact = 0
    for side in ("left", "right"):
        cols = [g1["qpos_addr"]["%s_shoulder_%s_joint" % (side, p)] for p in ("roll", "pitch", "yaw")]
        for c, part in zip(cols, ("roll", "pitch", "yaw")):
            lo, hi = g1["range"]["%s_shoulder_%s_joint" % (side, part)]
            assert full[:, c].min() >= lo - 1e-12 and full[:, c].max() <= hi + 1e-12
            assert np.abs(np.diff(full[:, c])).max() <= R.VMX_SHOULDER * dt + 1e-9
        for f in range(len(full)):
            hr, hp, hy = (naive[f, hadr["%s_shoulder_%s" % (side, a)]] for a in "xyz")     # what the block read (quirk)
            a, b, c = full[f, cols[0]], full[f, cols[1]], full[f, cols[2]]
            want = R.euler_matrix_rxyz(hr, hp, hy)
            ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
            rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
            exact += np.abs(ry @ rx @ rz - want).max() < 1e-9
    assert exact >= 0.5 * 2 * len(full), exact


def test_written_clip_loads_as_direct_qpos(tmp_path):
    """The tool's output is a clip MocapDM can load back (mocap_v2.py:271-272 format), labels as the reference writes them."""
    out = R.retarget_motion_humanoid_to_unitree_g1("run", out_path=str(tmp_path / "unitree_g1_run.txt"))
    d = json.load(open(out))
    assert d["Format"] == "direct_qpos" and d["Labels"] == [str(s) for s in G["labels"]] and len(d["Frames"][0]) == 45
    with pytest.raises(FileExistsError):
        R.retarget_motion_humanoid_to_unitree_g1("run", out_path=out)
