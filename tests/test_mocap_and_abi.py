"""CPU tests: mocap loader behaviour, package surfaces, C-ABI exports (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clip_lengths_and_singularity_fix_counts(clips):
    # L = 76 / 48 / 153 / 78 and the continuity fix fires 1 / 17 / 51 / 97 times
    assert {k: len(v.data_config) for k, v in clips.items()} == {"walk": 76, "run": 48, "dance_b": 153, "spinkick": 78}
    assert {k: v.singularity_fired for k, v in clips.items()} == {"walk": 1, "run": 17, "dance_b": 51, "spinkick": 97}


def test_clip_tables_are_consistent(model, clips):
    from deepmimic_mujoco_amd.model import forward_kinematics
    for name, mc in clips.items():
        q, v, b, g = mc.tables()
        assert q.shape[1] == 35 and v.shape[1] == 34 and b.shape[1:] == (14, 3) and g.shape[1:] == (16, 3)
        assert not v[0].any()                                  # frame 0 velocity is zero (mocap_v2.py:278-279)
        step = 2 if name in ("walk", "run") else 1             # raw (non-lerped) frames
        for i in range(0, len(q), step * 5):
            kin = forward_kinematics(model, q[i])
            assert np.abs(kin["geom_xpos"] - g[i]).max() < 1e-12   # intent of deepmimic_env.py:540-554
        # joint angles stay inside the model's ranges (the purpose of the singularity fix)
        lo, hi = model.jnt_range[1:, 0], model.jnt_range[1:, 1]
        viol = ((q[:, 7:] < lo - 0.2) | (q[:, 7:] > hi + 0.2)).mean()
        assert viol < 0.02


def test_interpolation_and_invalid_dt(tmp_path, model):
    import json
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.mocap import MocapDM
    data = json.load(open(MotionConfig("walk").mocap_path))
    for fr in data["Frames"]:
        fr[0] = 0.0625                                          # humanoid3d_backflip's dt: ratio 3.75 -> must raise
    p = tmp_path / "humanoid3d_bad.txt"
    p.write_text(json.dumps(data))
    with pytest.raises(Exception, match="Invalid dt ratio"):
        MocapDM(model=model).load_mocap(str(p))


def test_rot_vel_is_body_frame_angular_velocity():
    from deepmimic_mujoco_amd.mocap import calc_rot_vel
    from deepmimic_mujoco_amd.model import axis_angle_quat, quat_mul
    q0 = axis_angle_quat(np.array([0, 0, 1.0]), 0.7)
    dq = axis_angle_quat(np.array([1.0, 0, 0]), 0.02)
    w = calc_rot_vel(q0, quat_mul(q0, dq), 0.01)
    assert np.allclose(w, [2.0, 0, 0], atol=1e-9)
    assert calc_rot_vel(q0, q0, 0.01) == [0.0, 0.0, 0.0]


def test_config_mirrors_reference_names():
    from deepmimic_mujoco_amd.config import MotionConfig, RobotConfig
    rc = RobotConfig()
    assert rc.torso_body_name == "chest" and rc.low_z == 0.7
    assert rc.endeffector_geom_names == ["left_ankle", "right_ankle", "left_wrist", "right_wrist"]
    mc = MotionConfig()
    assert mc.motion == "walk" and os.path.exists(mc.mocap_path) and os.path.exists(mc.xml_path)
    assert "getup_facedown" in mc.floor_motions and "walk" not in mc.acyclical_motions
    g1 = RobotConfig("unitree_g1")                       # names for the retargeting tool; the physics refuses the model
    assert g1.torso_body_name == "pelvis" and g1.low_z == 0.4 and len(g1.extra_contact_geom_names) == 8
    assert os.path.exists(g1.xml_path)
    from deepmimic_mujoco_amd.model import compile_mjcf
    with pytest.raises(NotImplementedError):
        compile_mjcf(g1.xml_path)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from deepmimic_mujoco_amd import _lib
    L = _lib.load_library()
    hdr = open(os.path.join(ROOT, "include", "deepmimic_hip.h")).read()
    declared = set(re.findall(r"\b(dm_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    cfg = _lib.default_config()
    assert cfg.max_ep_length == 1000 and abs(cfg.vel_obs_scale - 0.1) < 1e-7 and abs(cfg.w_joint_limit + 0.1) < 1e-7


def test_g1_c_abi_exports_every_declared_symbol_and_the_struct_layout_matches():
    """include/deepmimic_g1_hip.h: every dmg1_* entry point is exported and the ctypes mirror of the G1 DmModel has the
    size the library was compiled with (no GPU call)."""
    import ctypes as C
    from deepmimic_mujoco_amd import _lib, g1
    L = _lib.load_library()
    hdr = open(os.path.join(ROOT, "include", "deepmimic_g1_hip.h")).read()
    declared = set(re.findall(r"\b(dmg1_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(g1.EXPORTS), declared ^ set(g1.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    L.dmg1_model_sizeof.restype = C.c_size_t
    assert L.dmg1_model_sizeof() == C.sizeof(g1.DmModelG1)
    cfg = g1.DmG1Config()
    L.dmg1_default_config.argtypes = [C.POINTER(g1.DmG1Config)]
    L.dmg1_default_config.restype = None
    L.dmg1_default_config(C.byref(cfg))
    assert cfg.max_ep_length == 1000 and abs(cfg.high_z - 2.0) < 1e-7
    gm, cm = g1.load_g1_model()
    assert cm.nq == 44 and cm.low_z == 0.4 and cm.action_scale == 20.0 and cm.n_policy_action == 23


def test_g1_engine_fails_loudly_without_gpu():
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G1HipEngine(4)


def test_engine_fails_loudly_without_gpu(model):
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEngine(model, 4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deepmimic_mujoco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "dm_oracle.h" not in src, f


def test_generated_topology_header_matches_the_xml(model):
    """csrc/dm_topology.h (constexpr dof tree the kernels are specialised to) must be in sync with the asset."""
    from deepmimic_mujoco_amd import gen_topology
    assert open(gen_topology.OUT).read() == gen_topology.render(model)
    parent = list(model.dof_parent)
    assert parent[:7] == [-1, 0, 1, 2, 3, 4, 5] and parent[20] == 5 and parent[27] == 5 and parent[12] == 8


def test_all_humanoid3d_clips_of_the_reference(model):
    """Every humanoid3d clip the reference ships either loads or raises exactly as mocap_v2.py:313-316 does
    (backflip dt 0.0625 and spin dt 0.041667 are not integer multiples of the simulator step)."""
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.mocap import MocapDM
    expect = {"cartwheel": 164, "crawl": 177, "dance_a": 98, "getup_facedown": 183, "getup_faceup": 227, "jump": 107,
              "kick": 92, "punch": 128, "roll": 121}          # kick/punch: dt 0.0333 -> (n-1)*2 frames
    for name, L in expect.items():
        mc = MocapDM(model=model)
        mc.load_mocap(MotionConfig(name).mocap_path)
        assert len(mc.data_config) == L, (name, len(mc.data_config))
        q, v, b, g = mc.tables()
        assert np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(g).all()
    for name in ("backflip", "spin"):
        with pytest.raises(Exception, match="Invalid dt ratio"):
            MocapDM(model=model).load_mocap(MotionConfig(name).mocap_path)
    cfg = MotionConfig("getup_facedown")
    assert cfg.motion in cfg.floor_motions and cfg.motion in cfg.acyclical_motions


def test_direct_qpos_format_round_trip(tmp_path, model, clips):
    """mocap_v2.py:271-272: a "Format": "direct_qpos" file carries qpos rows verbatim."""
    import json
    from deepmimic_mujoco_amd.mocap import MocapDM
    src = clips["spinkick"]                       # dt 0.016666: no interpolation, so tables must match exactly
    frames = [[src.dt] + list(map(float, q)) for q in src.data_config]
    p = tmp_path / "humanoid3d_direct.txt"
    p.write_text(json.dumps({"Loop": "wrap", "Format": "direct_qpos", "Frames": frames}))
    mc = MocapDM(model=model)
    mc.load_mocap(str(p))
    q0, v0, b0, g0 = src.tables()
    q1, v1, b1, g1 = mc.tables()
    assert np.array_equal(q0, q1) and np.allclose(v0, v1) and np.allclose(g0, g1) and np.allclose(b0, b1)
    bad = tmp_path / "bad.txt"
    bad.write_text(json.dumps({"Format": "direct_qpos", "Frames": [[0.0166] + [0.0] * 44]}))
    with pytest.raises(NotImplementedError):
        MocapDM(model=model).load_mocap(str(bad))       # 44-wide rows are the Unitree G1's qpos (out of scope)
