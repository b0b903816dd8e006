"""General MJCF compiler (deepmimic_mujoco_amd/mjcf.py): pinned on humanoid3d by model.py, then the Unitree G1 constants
(deepmimic_unitree_g1.xml) and the G1 direct_qpos clips through MocapDM(robot="unitree_g1")."""
import os

import numpy as np
import pytest

from deepmimic_mujoco_amd import mjcf, model as M
from deepmimic_mujoco_amd.config import MotionConfig, RobotConfig
from deepmimic_mujoco_amd.mocap import MocapDM

H3D = os.path.join(M.ASSET_DIR, "deepmimic_humanoid3d.xml")
G1 = os.path.join(M.ASSET_DIR, "deepmimic_unitree_g1.xml")


def test_humanoid3d_compiles_to_the_same_model_as_the_specialised_compiler():
    a, g = M.compile_mjcf(H3D), mjcf.compile_mjcf_general(H3D)
    for k in ["body_parent", "body_pos", "body_quat", "body_jntadr", "body_jntnum", "jnt_type", "jnt_body", "jnt_qposadr",
              "jnt_dofadr", "jnt_limited", "jnt_pos", "jnt_axis", "jnt_range", "dof_body", "dof_jnt", "dof_armature",
              "dof_damping", "dof_parent", "dof_Madr", "qpos0", "geom_type", "geom_body", "geom_condim", "geom_pos",
              "geom_quat", "geom_size", "geom_friction", "geom_margin", "geom_rbound", "body_mass", "body_ipos",
              "body_inertia", "act_dof", "act_gear", "act_ctrlrange"]:
        x, y = np.asarray(getattr(a, k), float), np.asarray(getattr(g, k), float)
        assert x.shape == y.shape and np.array_equal(x, y), k
    assert g.npair == a.npair == 104
    assert np.array_equal(g.pairs[:, 0], a.pair_geom1[:104]) and np.array_equal(g.pairs[:, 1], a.pair_geom2[:104])
    rng = np.random.default_rng(3)
    for _ in range(5):
        q = a.qpos0.copy()
        q[:3] += rng.normal(size=3)
        q[3:7] = rng.normal(size=4)
        q[7:] = rng.uniform(-1.2, 1.2, 28)
        k1, k2 = M.forward_kinematics(a, q), mjcf.forward_kinematics_general(g, q)
        for key in ["xpos", "xquat", "xmat", "xipos", "xaxis", "xanchor", "geom_xpos", "geom_xmat", "com"]:
            assert np.allclose(k1[key], k2[key], atol=1e-14), key


@pytest.fixture(scope="module")
def g1():
    return mjcf.compile_mjcf_general(G1, hulls=mjcf.load_g1_hulls())


def test_g1_dimensions_and_defaults(g1):
    # xml :7-10 RK4 / PGS / 50 iterations / h 0.0166 / nconmax 200; 1 free + 37 hinges
    assert (g1.nq, g1.nv, g1.nu, g1.nbody, g1.njnt) == (44, 43, 37, 39, 38)
    assert (g1.integrator, g1.solver, g1.iterations, g1.timestep, g1.nconmax) == ("RK4", "PGS", 50, 0.0166, 200)
    assert g1.jnt_type[0] == mjcf.JNT_FREE and (g1.jnt_type[1:] == mjcf.JNT_HINGE).all()
    # class g1 joint defaults (xml :16) reach every hinge through childclass, none reach the <freejoint>
    assert (g1.dof_damping[:6] == 0).all() and (g1.dof_armature[:6] == 0).all() and (g1.dof_frictionloss[:6] == 0).all()
    assert (g1.dof_damping[6:] == 0.5).all() and (g1.dof_armature[6:] == 0.01).all() and (g1.dof_frictionloss[6:] == 0.1).all()
    assert g1.jnt_limited[1:].all() and not g1.jnt_limited[0]
    # motors: 12 leg joints + waist with explicit ranges, arm class +-20, hand class +-0.7 (xml :17-22)
    hi = g1.act_ctrlrange[:, 1]
    assert np.array_equal(g1.act_ctrlrange[:, 0], -hi)
    assert list(hi[:13]) == [88, 88, 88, 139, 40, 40, 88, 88, 88, 139, 40, 40, 88]
    assert (hi[13:23] == 20).all() and (hi[23:] == 0.7).all()
    # actuator order is legs, waist, left arm, right arm, then the two hands — not the joint order (hands sit between the arms)
    assert sorted(g1.act_dof) == list(range(6, 43)) and list(g1.act_dof[:18]) == list(range(6, 24))
    assert list(g1.act_dof[18:23]) == list(range(31, 36))
    hand = [n for n in g1.act_names[23:]]
    assert len(hand) == 14 and all(("zero" in n or "one" in n or "two" in n or "three" in n or "four" in n or "five" in n
                                    or "six" in n) for n in hand), hand
    # the policy drives 23 of them; the 14 hand motors are padded with zeros (src/deepmimic_env.py:303-307, :348-351)
    assert g1.nu - 14 == 23
    assert abs(g1.body_mass.sum() - 32.2389206) < 1e-6
    assert np.allclose(g1.qpos0[:7], [0, 0, 0.755, 1, 0, 0, 0]) and (g1.qpos0[7:] == 0).all()


def test_g1_geoms_and_collision_classes(g1):
    names = RobotConfig("unitree_g1")
    for n in [names.lfoot_geom_name, names.rfoot_geom_name, names.floor_geom_name] + names.extra_contact_geom_names + \
            names.endeffector_geom_names:
        assert n in g1.geom_names, n
    assert g1.body_names[1] == names.torso_body_name == "pelvis"
    # visual class: group 2, no collisions; collision class: group 3, MuJoCo's default contype = conaffinity = 1, condim 3
    vis, col = g1.geom_group == 2, g1.geom_group == 3
    assert vis.sum() == 47 and col.sum() == 46 and g1.geom_type[0] == mjcf.GEOM_TYPES["plane"]
    assert (g1.geom_contype[vis] == 0).all() and (g1.geom_conaffinity[vis] == 0).all()
    assert (g1.geom_contype[col] == 1).all() and (g1.geom_condim == 3).all()
    t = g1.geom_type[col]
    assert {k: int((t == v).sum()) for k, v in mjcf.GEOM_TYPES.items() if (t == v).any()} == \
        {"sphere": 8, "cylinder": 4, "box": 2, "mesh": 32}
    # quirk the build keeps: the reference's G1 "foot" geoms are VISUAL spheres (xml :117), so DPEnv's foot-contact
    # observation (src/deepmimic_env.py:89-117) can never fire for this robot; the 8 foot spheres of class "foot" do collide
    for n in ("left_foot", "right_foot"):
        assert g1.geom_contype[g1.geom_id(n)] == 0
    for n in names.extra_contact_geom_names:
        i = g1.geom_id(n)
        assert g1.geom_type[i] == mjcf.GEOM_TYPES["sphere"] and g1.geom_size[i, 0] == 0.001 and g1.geom_contype[i] == 1
    # every collision mesh has its hull, and the hull contains no interior point of itself (convex position)
    for i in np.nonzero(col & (g1.geom_type == mjcf.GEOM_TYPES["mesh"]))[0]:
        h = g1.geom_hull[i]
        assert h is not None and len(h) >= 4 and g1.geom_rbound[i] > 0
    # candidate pairs never join a parent with its child and never involve a visual geom
    for a, b in g1.pairs:
        ba, bb = g1.geom_body[a], g1.geom_body[b]
        assert ba != bb and (ba == 0 or (g1.body_parent[bb] != ba and g1.body_parent[ba] != bb))
        assert g1.geom_contype[a] and g1.geom_contype[b]


def test_g1_kinematics_of_the_retargeted_clips(g1):
    """The G1 clips were produced by the reference's retargeting tool so that the feet stand on the floor: the general
    FK must put the 8 foot spheres of the walk clip within a centimetre or two of z = 0 at their lowest."""
    mc = MocapDM(robot="unitree_g1")
    mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
    q, v, bx, gx = mc.tables()
    assert q.shape == (76, 44) and v.shape == (76, 43) and bx.shape == (76, 39, 3) and gx.shape == (76, 94, 3)
    feet = [g1.geom_id(n) for n in RobotConfig("unitree_g1").extra_contact_geom_names]
    low = gx[:, feet, 2].min(0)
    assert (np.abs(low) < 0.02).all(), low
    assert np.allclose(bx[:, 1], q[:, :3])                       # pelvis = free-joint position
    assert np.allclose(v[1, 6:], (q[1, 7:] - q[0, 7:]) / mc.dt)  # mocap_v2.py:274-289
    # same-body geoms are rigidly attached: distances between the four spheres of one foot are constant over the clip
    d = np.linalg.norm(gx[:, feet[0]] - gx[:, feet[3]], axis=1)
    assert np.ptp(d) < 1e-12 and abs(d[0] - np.hypot(0.19, 0.04)) < 1e-12


@pytest.mark.parametrize("motion", ["run", "getup_facedown", "getup_facedown_slow", "getup_facedown_slow_FSI",
                                    "getup_facedown_towalk"])
def test_g1_clips_load(motion):
    mc = MocapDM(robot="unitree_g1")
    mc.load_mocap(MotionConfig(motion, robot="unitree_g1").mocap_path)
    q = np.array(mc.data_config)
    assert q.shape[1] == 44 and np.allclose(np.linalg.norm(q[:, 3:7], axis=1), 1, atol=0.05)     # interpolated quaternions are not renormalised (mocap_v2.py:325)
    assert abs(mc.dt - 0.01666) < 1e-4


def test_generated_g1_topology_header_matches_the_xml(g1):
    """csrc/dm_g1_topology.h (the compile-time 43-dof tree the G1 kernel's row solves are unrolled over) is in sync with the asset."""
    from deepmimic_mujoco_amd import gen_topology
    assert open(gen_topology.OUT_G1).read() == gen_topology.render_g1(g1)
    assert list(g1.dof_parent[:7]) == [-1, 0, 1, 2, 3, 4, 5] and g1.dof_parent[12] == 5 and g1.dof_parent[18] == 5 and g1.nM == 434
