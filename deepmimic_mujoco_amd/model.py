"""Host-side MJCF compiler: deepmimic_humanoid3d.xml -> DmModel (include/dm_model.h).

In the reference the model is compiled by MuJoCo's own XML compiler when gym's
``MujocoEnv.__init__`` loads the asset (src/deepmimic_env.py:301).  MuJoCo is not
available here, so this module restates the compile step for the subset of MJCF
the humanoid asset uses (src/mujoco/humanoid_deepmimic/envs/asset/
deepmimic_humanoid3d.xml:1-157): one <default>, nested bodies, free + hinge
joints, sphere/capsule(fromto)/box/plane geoms with explicit ``mass``
(``inertiafromgeom``), <motor> actuators, and the parent-child collision filter.

The float64 numpy kinematics/inertia helpers at the bottom are *compile-time*
tools (MuJoCo's ``mj_setConst`` equivalent: ``invweight0``, ``meaninertia``) and
are also what the motion loader uses to build the per-frame FK tables at load
time (reference: src/mujoco/mocap_v2.py:292-307).  They are not on the step()
hot path; that lives in csrc/ (HIP).
"""
from __future__ import annotations

import ctypes as C
import os
import xml.etree.ElementTree as ET

import numpy as np

NQ, NV, NU, NBODY, NGEOM, NJNT, NM = 35, 34, 28, 14, 16, 29, 310
MAXPAIR, NOBS, NEE = 128, 67, 4
MAXCON, MAXROW = 32, 128

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX = 0, 2, 3, 6
JNT_FREE, JNT_HINGE = 0, 3
INT_EULER, INT_RK4 = 0, 1

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

_i32, _f64 = C.c_int32, C.c_double


class DmModel(C.Structure):
    """ctypes mirror of ``struct DmModel`` (include/dm_model.h) — keep in sync."""

    _fields_ = [
        ("nq", _i32), ("nv", _i32), ("nu", _i32), ("nbody", _i32),
        ("ngeom", _i32), ("njnt", _i32), ("npair", _i32), ("nM", _i32),
        ("integrator", _i32), ("iterations", _i32), ("pad0", _i32), ("pad1", _i32),
        ("timestep", _f64), ("tolerance", _f64), ("gravity", _f64 * 3),
        ("meaninertia", _f64), ("solref", _f64 * 2), ("solimp", _f64 * 5),
        ("qpos0", _f64 * NQ),
        ("body_parent", _i32 * NBODY), ("body_jntadr", _i32 * NBODY),
        ("body_jntnum", _i32 * NBODY), ("body_dofadr", _i32 * NBODY),
        ("body_dofnum", _i32 * NBODY), ("body_depth", _i32 * NBODY),
        ("body_pos", _f64 * 3 * NBODY), ("body_quat", _f64 * 4 * NBODY),
        ("body_ipos", _f64 * 3 * NBODY), ("body_inertia", _f64 * 6 * NBODY),
        ("body_mass", _f64 * NBODY), ("body_invweight0", _f64 * 2 * NBODY),
        ("jnt_type", _i32 * NJNT), ("jnt_body", _i32 * NJNT),
        ("jnt_qposadr", _i32 * NJNT), ("jnt_dofadr", _i32 * NJNT),
        ("jnt_limited", _i32 * NJNT),
        ("jnt_pos", _f64 * 3 * NJNT), ("jnt_axis", _f64 * 3 * NJNT),
        ("jnt_range", _f64 * 2 * NJNT),
        ("dof_body", _i32 * NV), ("dof_jnt", _i32 * NV), ("dof_parent", _i32 * NV),
        ("dof_Madr", _i32 * NV),
        ("dof_armature", _f64 * NV), ("dof_damping", _f64 * NV),
        ("dof_invweight0", _f64 * NV),
        ("geom_type", _i32 * NGEOM), ("geom_body", _i32 * NGEOM),
        ("geom_condim", _i32 * NGEOM),
        ("geom_pos", _f64 * 3 * NGEOM), ("geom_quat", _f64 * 4 * NGEOM),
        ("geom_size", _f64 * 3 * NGEOM), ("geom_friction", _f64 * 3 * NGEOM),
        ("geom_margin", _f64 * NGEOM), ("geom_rbound", _f64 * NGEOM),
        ("act_dof", _i32 * NU), ("act_gear", _f64 * NU),
        ("act_ctrlrange", _f64 * 2 * NU),
        ("pair_geom1", _i32 * MAXPAIR), ("pair_geom2", _i32 * MAXPAIR),
        ("ee_geom", _i32 * NEE), ("torso_body", _i32),
        ("rfoot_geom", _i32), ("lfoot_geom", _i32), ("floor_geom", _i32),
    ]


# --------------------------------------------------------------------------
# small rotation helpers (wxyz quaternions, MuJoCo convention [EXT])
# --------------------------------------------------------------------------
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def axis_angle_quat(axis, angle):
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def z_to_quat(vec):
    """Quaternion rotating +z onto ``vec`` (MuJoCo compiler's fromto rule [EXT])."""
    vec = np.asarray(vec, float)
    vec = vec / np.linalg.norm(vec)
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = np.arctan2(s, vec[2])
    return axis_angle_quat(axis, ang)


def _floats(s):
    return [float(t) for t in s.split()]


# --------------------------------------------------------------------------
# MJCF parsing
# --------------------------------------------------------------------------
class CompiledModel:
    """Python-side view of the compiled model: numpy arrays + ``.cstruct``."""

    def __init__(self):
        self.body_names, self.geom_names, self.jnt_names, self.act_names = [], [], [], []


def _geom_inertia(gtype, size, mass):
    """(mass, diagonal inertia in the geom frame) — `inertiafromgeom` [EXT].

    sphere 2/5 m r^2; box m/3 (b^2+c^2) on half sizes; capsule = cylinder +
    two hemispheres with the mass split by volume.
    """
    if gtype == GEOM_SPHERE:
        r = size[0]
        i = 0.4 * mass * r * r
        return np.array([i, i, i])
    if gtype == GEOM_BOX:
        a, b, c = size
        return mass / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_CAPSULE:
        r, hl = size[0], size[1]
        height = 2.0 * hl
        vcyl = np.pi * r * r * height
        vsph = 4.0 / 3.0 * np.pi * r ** 3
        msph = mass * vsph / (vcyl + vsph)
        mcyl = mass - msph
        ixx = mcyl * (3 * r * r + height * height) / 12.0
        izz = mcyl * r * r / 2.0
        isph = 0.4 * msph * r * r
        ixx += isph + msph * height * (3 * r + 2 * height) / 8.0
        izz += isph
        return np.array([ixx, ixx, izz])
    raise ValueError("no inertia for geom type %d" % gtype)


def compile_mjcf(xml_path: str | None = None) -> CompiledModel:
    if xml_path is None:
        xml_path = os.path.join(ASSET_DIR, "deepmimic_humanoid3d.xml")
    root = ET.parse(xml_path).getroot()

    comp = root.find("compiler")
    if comp is None or comp.get("angle") != "radian" or comp.get("inertiafromgeom") != "true":
        raise NotImplementedError("unsupported <compiler>: the MJCF subset built here is the humanoid3d asset's (angle=radian, "
                                  "inertiafromgeom=true, primitive geoms); the Unitree G1 model (explicit inertials, mesh "
                                  "geoms, frictionloss) is SURVEY §8f-2, not built")

    dflt = {"joint": {}, "geom": {}, "motor": {}}
    d = root.find("default")
    if d is not None:
        for k in dflt:
            e = d.find(k)
            if e is not None:
                dflt[k] = dict(e.attrib)

    opt = root.find("option")
    oa = dict(opt.attrib) if opt is not None else {}

    m = CompiledModel()
    m.timestep = float(oa.get("timestep", 0.002))
    m.integrator = {"Euler": INT_EULER, "RK4": INT_RK4}[oa.get("integrator", "Euler")]
    if oa.get("solver", "Newton") != "PGS":
        raise ValueError("only solver=PGS is supported (reference xml :9)")
    m.iterations = int(oa.get("iterations", 100))
    m.tolerance = 1e-8
    m.gravity = np.array([0.0, 0.0, -9.81])
    m.solref = np.array([0.02, 1.0])
    m.solimp = np.array([0.9, 0.95, 0.001, 0.5, 2.0])

    bodies, joints, geoms = [], [], []

    def attr(e, kind, name, default=None):
        if name in e.attrib:
            return e.attrib[name]
        return dflt[kind].get(name, default)

    def add_geom(e, body_id):
        gtype = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
                 "box": GEOM_BOX}[e.get("type", "sphere")]
        size = _floats(e.get("size", "0"))
        pos = np.array(_floats(e.get("pos", "0 0 0")))
        quat = np.array([1.0, 0.0, 0.0, 0.0])
        if gtype == GEOM_CAPSULE:
            if "fromto" not in e.attrib:
                raise ValueError("capsule geoms must use fromto")
            ft = np.array(_floats(e.get("fromto")))
            a, b = ft[:3], ft[3:]
            pos = 0.5 * (a + b)
            quat = z_to_quat(b - a)
            size = [size[0], 0.5 * np.linalg.norm(b - a), 0.0]
        size = (list(size) + [0.0, 0.0, 0.0])[:3]
        fr = _floats(attr(e, "geom", "friction", "1 0.005 0.0001"))
        fr = (fr + [0.005, 0.0001])[:3] if len(fr) < 3 else fr
        g = dict(name=e.get("name", ""), type=gtype, body=body_id, pos=pos, quat=quat,
                 size=np.array(size, float),
                 mass=float(e.get("mass", 0.0)),
                 condim=int(attr(e, "geom", "condim", 3)),
                 contype=int(attr(e, "geom", "contype", 1)),
                 conaffinity=int(attr(e, "geom", "conaffinity", 1)),
                 margin=float(attr(e, "geom", "margin", 0.0)),
                 friction=np.array(fr, float))
        if gtype == GEOM_SPHERE:
            g["rbound"] = size[0]
        elif gtype == GEOM_CAPSULE:
            g["rbound"] = size[0] + size[1]
        elif gtype == GEOM_BOX:
            g["rbound"] = float(np.linalg.norm(size))
        else:
            g["rbound"] = 0.0
        geoms.append(g)

    def walk(e, parent_id, depth):
        bid = len(bodies)
        b = dict(name=e.get("name", "world"), parent=parent_id, depth=depth,
                 pos=np.array(_floats(e.get("pos", "0 0 0"))),
                 quat=np.array(_floats(e.get("quat", "1 0 0 0"))),
                 jntadr=-1, jntnum=0)
        bodies.append(b)
        # MuJoCo orders a body's joints/geoms in document order
        for j in e.findall("joint"):
            jt = j.get("type", "hinge")
            if jt not in ("free", "hinge"):
                raise ValueError("unsupported joint type " + jt)
            if b["jntnum"] == 0:
                b["jntadr"] = len(joints)
            b["jntnum"] += 1
            lim = attr(j, "joint", "limited", "false") == "true"
            joints.append(dict(
                name=j.get("name", ""), type=JNT_FREE if jt == "free" else JNT_HINGE,
                body=bid, pos=np.array(_floats(j.get("pos", "0 0 0"))),
                axis=np.array(_floats(j.get("axis", "0 0 1"))),
                limited=lim and jt != "free",
                range=np.array(_floats(j.get("range", "0 0"))),
                armature=float(attr(j, "joint", "armature", 0.0)),
                damping=float(attr(j, "joint", "damping", 0.0))))
        for g in e.findall("geom"):
            add_geom(g, bid)
        for c in e.findall("body"):
            walk(c, bid, depth + 1)

    wb = root.find("worldbody")
    walk(wb, -1, 0)
    bodies[0]["parent"] = 0
    bodies[0]["name"] = "world"

    if (len(bodies), len(geoms), len(joints)) != (NBODY, NGEOM, NJNT):
        raise ValueError("model dims %s do not match the compiled-in humanoid3d dims"
                         % ((len(bodies), len(geoms), len(joints)),))

    m.body_names = [b["name"] for b in bodies]
    m.geom_names = [g["name"] for g in geoms]
    m.jnt_names = [j["name"] for j in joints]
    m.body_parent = np.array([b["parent"] for b in bodies], np.int32)
    m.body_depth = np.array([b["depth"] for b in bodies], np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_quat = np.array([b["quat"] for b in bodies])
    m.body_jntadr = np.array([b["jntadr"] for b in bodies], np.int32)
    m.body_jntnum = np.array([b["jntnum"] for b in bodies], np.int32)

    # joints -> qpos/dof addresses
    qadr = dadr = 0
    m.jnt_type = np.zeros(NJNT, np.int32)
    m.jnt_body = np.zeros(NJNT, np.int32)
    m.jnt_qposadr = np.zeros(NJNT, np.int32)
    m.jnt_dofadr = np.zeros(NJNT, np.int32)
    m.jnt_limited = np.zeros(NJNT, np.int32)
    m.jnt_pos = np.zeros((NJNT, 3))
    m.jnt_axis = np.zeros((NJNT, 3))
    m.jnt_range = np.zeros((NJNT, 2))
    dof_body, dof_jnt, dof_arm, dof_damp = [], [], [], []
    for i, j in enumerate(joints):
        m.jnt_type[i] = j["type"]
        m.jnt_body[i] = j["body"]
        m.jnt_qposadr[i] = qadr
        m.jnt_dofadr[i] = dadr
        m.jnt_limited[i] = int(j["limited"])
        m.jnt_pos[i] = j["pos"]
        ax = j["axis"]
        m.jnt_axis[i] = ax / np.linalg.norm(ax)
        m.jnt_range[i] = j["range"]
        nd = 6 if j["type"] == JNT_FREE else 1
        qadr += 7 if j["type"] == JNT_FREE else 1
        for _ in range(nd):
            dof_body.append(j["body"])
            dof_jnt.append(i)
            dof_arm.append(j["armature"])
            dof_damp.append(j["damping"])
        dadr += nd
    if (qadr, dadr) != (NQ, NV):
        raise ValueError("nq/nv mismatch: %d/%d" % (qadr, dadr))
    if np.abs(m.jnt_pos).max() != 0.0:
        raise ValueError("HIP kernels assume jnt pos = 0 (true for humanoid3d)")
    m.dof_body = np.array(dof_body, np.int32)
    m.dof_jnt = np.array(dof_jnt, np.int32)
    m.dof_armature = np.array(dof_arm)
    m.dof_damping = np.array(dof_damp)
    m.body_dofadr = np.full(NBODY, -1, np.int32)
    m.body_dofnum = np.zeros(NBODY, np.int32)
    for k in range(NV):
        b = m.dof_body[k]
        if m.body_dofnum[b] == 0:
            m.body_dofadr[b] = k
        m.body_dofnum[b] += 1
    # dof parent chain: previous dof of the same body, else last dof of the
    # nearest ancestor body that has dofs [EXT mjModel.dof_parentid]
    m.dof_parent = np.full(NV, -1, np.int32)
    for k in range(NV):
        b = m.dof_body[k]
        if k > m.body_dofadr[b]:
            m.dof_parent[k] = k - 1
        else:
            p = m.body_parent[b]
            while p > 0 and m.body_dofnum[p] == 0:
                p = m.body_parent[p]
            if p > 0:
                m.dof_parent[k] = m.body_dofadr[p] + m.body_dofnum[p] - 1
    m.dof_Madr = np.zeros(NV, np.int32)
    adr = 0
    for k in range(NV):
        m.dof_Madr[k] = adr
        j = k
        while j >= 0:
            adr += 1
            j = m.dof_parent[j]
    if adr != NM:
        raise ValueError("nM mismatch: %d" % adr)

    m.qpos0 = np.zeros(NQ)
    rb = joints[0]["body"]
    m.qpos0[0:3] = bodies[rb]["pos"]
    m.qpos0[3:7] = bodies[rb]["quat"]

    # geoms
    m.geom_type = np.array([g["type"] for g in geoms], np.int32)
    m.geom_body = np.array([g["body"] for g in geoms], np.int32)
    m.geom_condim = np.array([g["condim"] for g in geoms], np.int32)
    m.geom_pos = np.array([g["pos"] for g in geoms])
    m.geom_quat = np.array([g["quat"] for g in geoms])
    m.geom_size = np.array([g["size"] for g in geoms])
    m.geom_friction = np.array([g["friction"] for g in geoms])
    m.geom_margin = np.array([g["margin"] for g in geoms])
    m.geom_rbound = np.array([g["rbound"] for g in geoms])
    m.geom_mass = np.array([g["mass"] for g in geoms])

    # body inertial properties from geoms (parallel-axis merge about the COM)
    m.body_mass = np.zeros(NBODY)
    m.body_ipos = np.zeros((NBODY, 3))
    m.body_inertia = np.zeros((NBODY, 6))
    for b in range(1, NBODY):
        gs = [g for g in geoms if g["body"] == b and g["mass"] > 0]
        mass = sum(g["mass"] for g in gs)
        com = sum(g["mass"] * g["pos"] for g in gs) / mass
        I = np.zeros((3, 3))
        for g in gs:
            R = quat_to_mat(g["quat"])
            Ig = R @ np.diag(_geom_inertia(g["type"], g["size"], g["mass"])) @ R.T
            dlt = g["pos"] - com
            I += Ig + g["mass"] * (dlt @ dlt * np.eye(3) - np.outer(dlt, dlt))
        m.body_mass[b] = mass
        m.body_ipos[b] = com
        m.body_inertia[b] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    # actuators
    acts = root.find("actuator").findall("motor")
    if len(acts) != NU:
        raise ValueError("nu mismatch")
    m.act_names = [a.get("name") for a in acts]
    m.act_dof = np.zeros(NU, np.int32)
    m.act_gear = np.zeros(NU)
    m.act_ctrlrange = np.zeros((NU, 2))
    for i, a in enumerate(acts):
        jn = a.get("joint")
        ji = m.jnt_names.index(jn)
        m.act_dof[i] = m.jnt_dofadr[ji]
        m.act_gear[i] = float(a.get("gear", "1").split()[0])
        if attr(a, "motor", "ctrllimited", "false") == "true":
            m.act_ctrlrange[i] = _floats(attr(a, "motor", "ctrlrange", "0 0"))
        else:
            m.act_ctrlrange[i] = [-np.inf, np.inf]

    # collision candidates: different bodies, contype/conaffinity compatible,
    # not parent-child (bodies whose parent is the world may touch world geoms),
    # minus <exclude> (all redundant here).  Canonical order = (body1,body2)
    # ascending then geom order; lower geom type first inside a pair.
    excl = set()
    ct = root.find("contact")
    if ct is not None:
        for e in ct.findall("exclude"):
            b1 = m.body_names.index(e.get("body1"))
            b2 = m.body_names.index(e.get("body2"))
            excl.add((min(b1, b2), max(b1, b2)))
    pairs = []
    for b1 in range(NBODY):
        for b2 in range(b1 + 1, NBODY):
            if (b1, b2) in excl:
                continue
            if b1 != 0 and (m.body_parent[b2] == b1 or m.body_parent[b1] == b2):
                continue
            for g1 in [i for i in range(NGEOM) if m.geom_body[i] == b1]:
                for g2 in [i for i in range(NGEOM) if m.geom_body[i] == b2]:
                    ga, gb = geoms[g1], geoms[g2]
                    if not ((ga["contype"] & gb["conaffinity"]) or (gb["contype"] & ga["conaffinity"])):
                        continue
                    if ga["type"] > gb["type"]:
                        pairs.append((g2, g1))
                    else:
                        pairs.append((g1, g2))
    if len(pairs) > MAXPAIR:
        raise ValueError("too many collision pairs")
    m.npair = len(pairs)
    m.pair_geom1 = np.full(MAXPAIR, -1, np.int32)
    m.pair_geom2 = np.full(MAXPAIR, -1, np.int32)
    for i, (a, b) in enumerate(pairs):
        m.pair_geom1[i], m.pair_geom2[i] = a, b

    # task constants (src/config.py:6-13)
    m.ee_geom = np.array([m.geom_names.index(n) for n in
                          ["left_ankle", "right_ankle", "left_wrist", "right_wrist"]], np.int32)
    m.torso_body = m.body_names.index("chest")
    m.rfoot_geom = m.geom_names.index("right_ankle")
    m.lfoot_geom = m.geom_names.index("left_ankle")
    m.floor_geom = m.geom_names.index("floor")

    # constants that need M(qpos0): invweight0 / meaninertia [EXT mj_setConst]
    kin = forward_kinematics(m, m.qpos0)
    M = mass_matrix(m, kin)
    Minv = np.linalg.inv(M)
    m.meaninertia = float(np.mean(np.diag(M)))
    dw = np.diag(Minv).copy()
    dw[0:3] = dw[0:3].mean()
    dw[3:6] = dw[3:6].mean()
    m.dof_invweight0 = dw
    m.body_invweight0 = np.zeros((NBODY, 2))
    for b in range(1, NBODY):
        jp, jr = jacobian(m, kin, kin["xipos"][b], b)
        A = np.vstack([jp, jr]) @ Minv @ np.vstack([jp, jr]).T
        m.body_invweight0[b, 0] = (A[0, 0] + A[1, 1] + A[2, 2]) / 3.0
        m.body_invweight0[b, 1] = (A[3, 3] + A[4, 4] + A[5, 5]) / 3.0

    m.cstruct = _to_cstruct(m)
    return m


def _to_cstruct(m: CompiledModel) -> DmModel:
    s = DmModel()
    s.nq, s.nv, s.nu, s.nbody, s.ngeom, s.njnt, s.npair, s.nM = NQ, NV, NU, NBODY, NGEOM, NJNT, m.npair, NM
    s.integrator, s.iterations = m.integrator, m.iterations
    s.timestep, s.tolerance, s.meaninertia = m.timestep, m.tolerance, m.meaninertia
    s.torso_body, s.rfoot_geom, s.lfoot_geom, s.floor_geom = (
        int(m.torso_body), int(m.rfoot_geom), int(m.lfoot_geom), int(m.floor_geom))

    def put(name, arr):
        field = getattr(s, name)
        a = np.ascontiguousarray(arr)
        dst = np.ctypeslib.as_array(field)
        dst[...] = a.reshape(dst.shape)

    for name in ["gravity", "solref", "solimp", "qpos0", "body_parent", "body_jntadr",
                 "body_jntnum", "body_dofadr", "body_dofnum", "body_depth", "body_pos",
                 "body_quat", "body_ipos", "body_inertia", "body_mass", "body_invweight0",
                 "jnt_type", "jnt_body", "jnt_qposadr", "jnt_dofadr", "jnt_limited",
                 "jnt_pos", "jnt_axis", "jnt_range", "dof_body", "dof_jnt", "dof_parent",
                 "dof_Madr", "dof_armature", "dof_damping", "dof_invweight0", "geom_type",
                 "geom_body", "geom_condim", "geom_pos", "geom_quat", "geom_size",
                 "geom_friction", "geom_margin", "geom_rbound", "act_dof", "act_gear",
                 "act_ctrlrange", "pair_geom1", "pair_geom2", "ee_geom"]:
        put(name, getattr(m, name))
    return s


# --------------------------------------------------------------------------
# compile-time / load-time float64 kinematics (NOT the step() hot path)
# --------------------------------------------------------------------------
def forward_kinematics(m: CompiledModel, qpos):
    """Body/geom world poses for one configuration (MuJoCo mj_kinematics [EXT]).

    Used at compile time (invweight0) and at clip-load time for the per-frame
    ``body_xpos``/``geom_xpos`` tables (src/mujoco/mocap_v2.py:301-307 does this
    with a mocap-less DPEnv + set_state)."""
    qpos = np.asarray(qpos, float)
    xpos = np.zeros((NBODY, 3))
    xquat = np.zeros((NBODY, 4))
    xquat[0] = [1, 0, 0, 0]
    xmat = np.zeros((NBODY, 3, 3))
    xmat[0] = np.eye(3)
    xaxis = np.zeros((NJNT, 3))
    xanchor = np.zeros((NJNT, 3))
    for b in range(1, NBODY):
        p = m.body_parent[b]
        ja, jn = m.body_jntadr[b], m.body_jntnum[b]
        if jn == 1 and m.jnt_type[ja] == JNT_FREE:
            pos = qpos[0:3].copy()
            q = qpos[3:7] / np.linalg.norm(qpos[3:7])
            xanchor[ja] = pos
            xaxis[ja] = quat_to_mat(q) @ m.jnt_axis[ja]
        else:
            pos = xpos[p] + xmat[p] @ m.body_pos[b]
            q = quat_mul(xquat[p], m.body_quat[b])
            for j in range(ja, ja + jn):
                R = quat_to_mat(q)
                xanchor[j] = R @ m.jnt_pos[j] + pos
                xaxis[j] = R @ m.jnt_axis[j]
                ang = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                q = quat_mul(q, axis_angle_quat(m.jnt_axis[j], ang))
                pos = xanchor[j] - quat_to_mat(q) @ m.jnt_pos[j]
        q = q / np.linalg.norm(q)
        xpos[b], xquat[b], xmat[b] = pos, q, quat_to_mat(q)
    xipos = xpos + np.einsum("bij,bj->bi", xmat, m.body_ipos)
    gb = m.geom_body
    geom_xpos = xpos[gb] + np.einsum("gij,gj->gi", xmat[gb], m.geom_pos)
    geom_xmat = np.array([xmat[gb[g]] @ quat_to_mat(m.geom_quat[g]) for g in range(NGEOM)])
    mtot = m.body_mass.sum()
    com = (m.body_mass[:, None] * xipos).sum(0) / mtot
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, xaxis=xaxis,
                xanchor=xanchor, geom_xpos=geom_xpos, geom_xmat=geom_xmat, com=com)


def jacobian(m: CompiledModel, kin, point, body):
    """Translational/rotational Jacobian (3 x nv each) of ``point`` fixed to ``body``."""
    jp = np.zeros((3, NV))
    jr = np.zeros((3, NV))
    b = body
    chain = []
    while b > 0:
        chain.append(b)
        b = m.body_parent[b]
    for b in chain:
        for k in range(m.body_dofadr[b], m.body_dofadr[b] + m.body_dofnum[b]):
            j = m.dof_jnt[k]
            if m.jnt_type[j] == JNT_FREE:
                i = k - m.jnt_dofadr[j]
                if i < 3:
                    jp[i, k] = 1.0
                else:
                    ax = kin["xmat"][b][:, i - 3]
                    jr[:, k] = ax
                    jp[:, k] = np.cross(ax, point - kin["xanchor"][j])
            else:
                ax = kin["xaxis"][j]
                jr[:, k] = ax
                jp[:, k] = np.cross(ax, point - kin["xanchor"][j])
    return jp, jr


def mass_matrix(m: CompiledModel, kin):
    """Dense joint-space inertia incl. armature: sum_b J_b^T diag(m, I_b) J_b."""
    M = np.zeros((NV, NV))
    for b in range(1, NBODY):
        jp, jr = jacobian(m, kin, kin["xipos"][b], b)
        xx, yy, zz, xy, xz, yz = m.body_inertia[b]
        Ib = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        Iw = kin["xmat"][b] @ Ib @ kin["xmat"][b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    M += np.diag(m.dof_armature)
    return M


_CACHE = {}


def load_model(xml_path: str | None = None) -> CompiledModel:
    key = xml_path or "humanoid3d"
    if key not in _CACHE:
        _CACHE[key] = compile_mjcf(xml_path)
    return _CACHE[key]
