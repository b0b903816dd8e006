"""Unitree G1 build of the fp64 oracle (oracle/libdm_oracle_g1.so): dm_oracle.c at the G1 dimensions with
convex narrowphase (libccd MPR restated), plane-cylinder / plane-mesh, friction-loss rows and the G1 branch of DPEnv.

Physics parity is unpinned (no MuJoCo here, no golden contacts in the reference tree): these tests pin the restatement to
its own analytic primitives, to geometric invariants and to Newton's second law, and the task layer to a numpy restatement of
src/deepmimic_env.py:193-256,335-476.
"""
import ctypes as C
import math

import numpy as np
import pytest

from deepmimic_mujoco_amd import mjcf
from deepmimic_mujoco_amd.config import MotionConfig, RobotConfig
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.model import quat_mul
from oracle import oracle_g1 as og

SPH, CAP, CYL, BOX, MESH = 2, 3, 5, 6, 7


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    L = og.lib()
    L.dmo_mpr.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    L.dmo_narrowphase.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_double, C.c_void_p]
    L.dmo_plane_mesh.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_double, C.c_void_p]
    return L


def _rot(rng):
    q = rng.normal(size=4)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _mpr(L, t1, x1, M1, z1, t2, x2, M2, z2, v1=None, v2=None):
    out = np.zeros(7)
    M1, M2 = np.ascontiguousarray(M1), np.ascontiguousarray(M2)
    n = L.dmo_mpr(t1, _p(x1), _p(M1), _p(z1), _p(v1) if v1 is not None else None, 0 if v1 is None else len(v1),
                  t2, _p(x2), _p(M2), _p(z2), _p(v2) if v2 is not None else None, 0 if v2 is None else len(v2), _p(out))
    return n, out


def _analytic(L, t1, x1, M1, z1, t2, x2, M2, z2, margin=0.0):
    out = np.zeros(80)
    M1, M2 = np.ascontiguousarray(M1), np.ascontiguousarray(M2)
    n = L.dmo_narrowphase(t1, _p(x1), _p(M1), _p(z1), t2, _p(x2), _p(M2), _p(z2), margin, _p(out))
    return n, out[:10 * max(n, 0)].reshape(-1, 10)


def _size(rng, t):
    if t == SPH:
        return np.array([rng.uniform(.05, .2), 0, 0])
    if t in (CAP, CYL):
        return np.array([rng.uniform(.03, .08), rng.uniform(.05, .2), 0])
    return rng.uniform(.05, .2, 3)


def _support(t, x, M, z, verts, d):
    dl = M.T @ d
    if t == SPH:
        p = dl / np.linalg.norm(dl) * z[0]
    elif t == CAP:
        p = dl / np.linalg.norm(dl) * z[0] + np.array([0, 0, math.copysign(z[1], dl[2])])
    elif t == CYL:
        n = math.hypot(dl[0], dl[1])
        p = np.array([dl[0] / n * z[0] if n > 0 else 0, dl[1] / n * z[0] if n > 0 else 0, math.copysign(z[1], dl[2])])
    elif t == BOX:
        p = np.sign(dl) * z
    else:
        p = verts[np.argmax(verts @ dl)]
    return M @ p + x


@pytest.mark.parametrize("t1,t2", [(SPH, SPH), (SPH, CAP), (SPH, BOX), (CAP, CAP), (CAP, BOX), (BOX, BOX)])
def test_mpr_agrees_with_the_analytic_primitives(lib, t1, t2):
    """Same intersect / separate verdict on every random pair; on shallow penetrations (< 5 mm) the MPR depth equals the
    analytic routine's (MPR is approximate by construction: its direction comes from the portal the centre ray hits)."""
    rng = np.random.default_rng(10 * t1 + t2)
    diffs, hits = [], 0
    for _ in range(1500):
        z1, z2, M1, M2 = _size(rng, t1), _size(rng, t2), _rot(rng), _rot(rng)
        x1, x2 = np.zeros(3), rng.normal(size=3) * 0.15
        na, ca = _analytic(lib, t1, x1, M1, z1, t2, x2, M2, z2)
        nm, cm = _mpr(lib, t1, x1, M1, z1, t2, x2, M2, z2)
        pen = [r for r in ca if r[0] < -1e-9]
        touching = any(abs(r[0]) <= 1e-9 for r in ca)
        if not touching:
            assert bool(pen) == bool(nm), (t1, t2, ca, cm)
        if not pen or not nm:
            continue
        hits += 1
        d = min(r[0] for r in pen)
        if d > -0.005:
            diffs.append(abs(cm[0] - d))
            assert abs(np.linalg.norm(cm[4:7]) - 1) < 1e-12
    assert hits > 100 and len(diffs) >= 10
    assert max(diffs) < 6e-3 and np.median(diffs) < 2e-5, (max(diffs), np.median(diffs))


def _minkowski_depth(pa, pb):
    """Exact signed penetration depth of two polytopes given by world vertices: min facet offset of hull(A - B)
    (> 0: overlap depth, < 0: separated)."""
    from scipy.spatial import ConvexHull
    d = (pa[:, None, :] - pb[None, :, :]).reshape(-1, 3)
    hull = ConvexHull(d)
    return float((-hull.equations[:, 3]).min())       # equations: n.x + off <= 0 inside


def _box_vertices(x, M, z):
    c = np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], float) * z
    return c @ M.T + x


@pytest.mark.parametrize("t1,t2", [(MESH, BOX), (MESH, MESH), (BOX, BOX)])
def test_mpr_on_polytopes_against_the_exact_minkowski_depth(lib, t1, t2):
    """Polytope pairs have an exact answer (facets of hull(A - B)): MPR must give the same intersect / separate verdict, never
    report less than the true depth (its converged portal is a supporting plane of A - B), and its (depth, normal) must stay
    inside A - B: depth <= h(normal)."""
    g, _ = og.g1_model()
    rng = np.random.default_rng(100 * t1 + t2)
    hull = np.ascontiguousarray(g.mesh_vert[g.mesh_names.index("left_ankle_pitch_link")], np.float64)   # 129 vertices
    found, shallow = 0, []
    for _ in range(120):
        v1, v2 = (hull if t1 == MESH else None), (hull if t2 == MESH else None)
        z1, z2, M1, M2 = _size(rng, BOX) * 0.3, _size(rng, BOX) * 0.3, _rot(rng), _rot(rng)
        x1 = -(M1 @ hull.mean(0)) if t1 == MESH else np.zeros(3)
        x2 = rng.normal(size=3) * 0.03 - (M2 @ hull.mean(0) if t2 == MESH else 0)
        pa = hull @ M1.T + x1 if t1 == MESH else _box_vertices(x1, M1, z1)
        pb = hull @ M2.T + x2 if t2 == MESH else _box_vertices(x2, M2, z2)
        true = _minkowski_depth(pa, pb)
        n, c = _mpr(lib, t1, x1, M1, z1, t2, x2, M2, z2, v1, v2)
        if abs(true) < 1e-7:
            continue
        assert bool(n) == (true > 0), (true, n, c)
        if not n:
            continue
        found += 1
        depth, nrm = -c[0], c[4:7]
        hn = float((pa @ nrm).max() - (pb @ nrm).min())
        assert abs(np.linalg.norm(nrm) - 1) < 1e-9
        assert true - 1e-6 <= depth <= hn + 1e-5, (true, depth, hn)
        if true < 0.002:
            shallow.append(depth - true)
    assert found > 15
    if shallow:
        assert np.median(shallow) < 1e-4


@pytest.mark.parametrize("t1,t2", [(CYL, CYL), (CYL, BOX), (SPH, CYL), (MESH, CYL)])
def test_mpr_with_cylinders_certified_verdicts(lib, t1, t2):
    """Cylinder pairs (no exact reference): a direction with h(d) < 0 certifies separation, a common point certifies overlap;
    MPR must agree whenever either certificate exists, and keep depth <= h(normal)."""
    g, _ = og.g1_model()
    rng = np.random.default_rng(100 * t1 + t2)
    hull = np.ascontiguousarray(g.mesh_vert[g.mesh_names.index("left_ankle_pitch_link")], np.float64)
    dirs = rng.normal(size=(3000, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)

    def inside(t, x, M, z, p):
        pl = (p - x) @ M                      # world -> local (rows)
        if t == SPH:
            return np.linalg.norm(pl, axis=1) <= z[0]
        if t == CYL:
            return (np.hypot(pl[:, 0], pl[:, 1]) <= z[0]) & (np.abs(pl[:, 2]) <= z[1])
        return (np.abs(pl) <= z).all(1)

    def sample(t, x, M, z, n):
        if t == MESH:
            w = rng.dirichlet(np.ones(6), size=n)
            idx = rng.integers(len(hull), size=(n, 6))
            return (w[:, :, None] * hull[idx]).sum(1) @ M.T + x
        if t == SPH:
            p = rng.normal(size=(n, 3))
            p = p / np.linalg.norm(p, axis=1, keepdims=True) * z[0] * rng.uniform(0, 1, (n, 1)) ** (1 / 3)
        elif t == CYL:
            r, th = z[0] * np.sqrt(rng.uniform(0, 1, n)), rng.uniform(0, 2 * np.pi, n)
            p = np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-z[1], z[1], n)], 1)
        else:
            p = rng.uniform(-1, 1, (n, 3)) * z
        return p @ M.T + x

    sep = ovl = 0
    for _ in range(150):
        v1 = hull if t1 == MESH else None
        z1, z2, M1, M2 = _size(rng, t1) * (1 if t1 != MESH else 0), _size(rng, t2), _rot(rng), _rot(rng)
        x1 = -(M1 @ hull.mean(0)) if t1 == MESH else np.zeros(3)
        x2 = rng.normal(size=3) * (0.06 if t1 == MESH else 0.12)
        n, c = _mpr(lib, t1, x1, M1, z1, t2, x2, M2, z2, v1, None)
        hs = np.array([(_support(t1, x1, M1, z1, v1, d) - _support(t2, x2, M2, z2, None, -d)) @ d for d in dirs[:600]])
        if hs.min() < -1e-9:
            sep += 1
            assert n == 0
            continue
        pts = sample(t1, x1, M1, z1, 4000)
        if inside(t2, x2, M2, z2, pts).any():
            ovl += 1
            assert n == 1
        if n:
            depth, nrm = -c[0], c[4:7]
            assert depth >= 0 and abs(np.linalg.norm(nrm) - 1) < 1e-9
            if SPH not in (t1, t2):           # spheres get the analytic normal afterwards (mjc_fixNormal)
                hn = float((_support(t1, x1, M1, z1, v1, nrm) - _support(t2, x2, M2, z2, None, -nrm)) @ nrm)
                assert depth <= hn + 1e-5
    assert sep > 10 and ovl > 10


def test_plane_cylinder_against_rim_sampling(lib):
    """mjc_PlaneCylinder: contact 0 is the deepest rim point; every contact lies half its depth above a point of the rim."""
    rng = np.random.default_rng(5)
    ppos, pmat, zero = np.zeros(3), np.eye(3), np.zeros(3)
    th = np.linspace(0, 2 * np.pi, 7201)[:-1]
    seen = {}
    for _ in range(300):
        z, M = _size(rng, CYL), _rot(rng)
        x = np.array([0, 0, rng.uniform(0.0, 0.25)])
        rim = np.concatenate([np.stack([z[0] * np.cos(th), z[0] * np.sin(th), np.full_like(th, s * z[1])], 1) for s in (-1, 1)])
        wr = rim @ M.T + x
        n, c = _analytic(lib, 0, ppos, pmat, zero, CYL, x, M, z)
        if wr[:, 2].min() > 1e-6:
            assert n == 0
            continue
        assert n >= 1
        seen[n] = seen.get(n, 0) + 1
        assert abs(c[0, 0] - wr[:, 2].min()) < 1e-6
        for r in c:
            assert np.allclose(r[4:7], [0, 0, 1]) and r[0] <= 1e-12
            on_surface = r[1:4] + np.array([0, 0, 0.5 * r[0]])
            assert np.min(np.linalg.norm(wr - on_surface, axis=1)) < z[0] * 2e-3 + 1e-6
    assert set(seen) >= {1, 2}


def test_plane_mesh_first_contact_is_the_lowest_hull_vertex(lib):
    g, _ = og.g1_model()
    rng = np.random.default_rng(6)
    ppos, pmat = np.zeros(3), np.ascontiguousarray(np.eye(3))
    multi = 0
    for name in ("pelvis", "torso_link", "left_palm_link", "head_link"):
        v = np.ascontiguousarray(g.mesh_vert[g.mesh_names.index(name)], np.float64)
        for _ in range(20):
            M = np.ascontiguousarray(_rot(rng))
            w = v @ M.T
            x = np.array([0.3, -0.2, -w[:, 2].min() - rng.uniform(-0.002, 0.004)])
            out = np.zeros(28)
            n = lib.dmo_plane_mesh(_p(ppos), _p(pmat), _p(x), _p(M), _p(v), len(v), 0.0, _p(out))
            low = (w + x)[:, 2].min()
            assert (n > 0) == (low <= 0)
            c = out[:7 * n].reshape(-1, 7)
            if n:
                assert abs(c[0, 0] - low) < 1e-15 and n <= 4
                assert (c[:, 0] <= 0).all() and np.allclose(c[:, 4:7], [0, 0, 1])
                # every contact is a hull vertex lifted by half its depth; all distinct
                for r in c:
                    assert np.min(np.linalg.norm(w + x - (r[1:4] + [0, 0, 0.5 * r[0]]), axis=1)) < 1e-12
                assert len({tuple(np.round(r[1:4], 12)) for r in c}) == n
                multi += n > 1
    assert multi > 0


def test_plane_mesh_contacts_do_not_depend_on_the_vertex_order(lib):
    """the plane-mesh contact set (a restatement of MuJoCo >= 2.1's mjc_PlaneConvex multi-contact rule, from
    its documentation only: LOW CONFIDENCE) is a set of hull vertices within the margin that contains the deepest one (test above)
    and is a function of the geometry alone: the same contacts for any order of the vertex array (generic orientations: no ties)."""
    g, _ = og.g1_model()
    rng = np.random.default_rng(16)
    ppos, pmat = np.zeros(3), np.ascontiguousarray(np.eye(3))
    for name in ("pelvis", "torso_link", "left_palm_link"):
        v = np.ascontiguousarray(g.mesh_vert[g.mesh_names.index(name)], np.float64)
        for _ in range(10):
            M = np.ascontiguousarray(_rot(rng))
            x = np.array([0.1, 0.2, -(v @ M.T)[:, 2].min() - rng.uniform(0.0, 0.004)])
            sets = []
            for perm in (np.arange(len(v)), rng.permutation(len(v)), np.arange(len(v))[::-1]):
                vp, out = np.ascontiguousarray(v[perm]), np.zeros(28)
                n = lib.dmo_plane_mesh(_p(ppos), _p(pmat), _p(x), _p(M), _p(vp), len(vp), 0.0, _p(out))
                sets.append(sorted(tuple(np.round(r, 12)) for r in out[:7 * n].reshape(-1, 7)))
            assert sets[0] == sets[1] == sets[2] and len(sets[0]) >= 1


def test_mpr_support_tie_is_structural_and_the_tie_rule_removes_it(walk):
    """The finding behind r3's G1 parity (oracle/dm_convex.h "ties"): frame 14 of the walk clip has the right hand's
    finger hull 9 mm inside the hip hull (an edge-edge contact).  With libccd's literal support scan (support_tie = 0) the ORACLE
    ITSELF returns one of two contacts (depth 8.90 / 9.09 mm, normals 0.035 apart) depending on 1e-13 perturbations of the joint
    angles: when two portal vertices share a witness, two hull vertices tie exactly along the portal normal and the last bit of
    the dot products picks.  With the tie rule (the default) the contact is a function of the pose again."""
    mc, _ = walk
    q0 = np.asarray(mc.data_config[14], np.float64).astype(np.float32).astype(np.float64)
    v0 = np.asarray(mc.data_vel[14], np.float64).astype(np.float32).astype(np.float64)
    L = og.lib()

    def normals(tie):
        assert L.dmo_set_tweak(b"support_tie", tie) == 0
        rng = np.random.default_rng(0)
        out = []
        for k in range(16):
            q = q0.copy()
            q[7:] += rng.normal(size=37) * (1e-13 if k else 0.0)
            s = og.G1Sim()
            s.set_caps(48, 256)
            assert s.set_state(q, v0) == 0
            c = [c for c in s.contacts() if (c["geom1"], c["geom2"]) == (23, 85)]
            assert len(c) == 1
            out.append(np.concatenate([[c[0]["dist"]], c[0]["frame"][0]]))
        return np.array(out)

    try:
        lit = normals(0.0)
        rule = normals(1e-12)
    finally:
        L.dmo_set_tweak(b"reset", 0.0)
    spread_lit = np.abs(lit - lit[0]).max(axis=1)
    print("literal scan: distinct contacts", len({tuple(np.round(r, 6)) for r in lit}), "max spread", spread_lit.max(),
          "| tie rule: max spread", np.abs(rule - rule[0]).max())
    assert (spread_lit > 1e-2).sum() >= 3 and (spread_lit < 1e-9).sum() >= 3      # two answers, both frequent
    assert np.abs(rule - rule[0]).max() < 1e-9                                    # one answer
    assert min(np.abs(lit - rule[0]).max(axis=1)) < 1e-9                          # ... which is one of the literal scan's two


# ------------------------------------------------------------------------------------------ dynamics
def _comvel(g, q, v):
    k = mjcf.forward_kinematics_general(g, q)
    J = np.zeros((3, g.nv))
    for b in range(1, g.nbody):
        J += g.body_mass[b] * mjcf.jacobian_general(g, k, k["xipos"][b], b)[0]
    return J @ v / g.body_mass.sum()


def _integ(q, v, h):
    q2 = q.copy()
    q2[:3] += h * v[:3]
    w = v[3:6]
    ang = np.linalg.norm(w) * h
    if ang != 0:
        ax = w / np.linalg.norm(w)
        q2[3:7] = quat_mul(q[3:7], np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax]))
    q2[7:] += h * v[6:]
    return q2


def _com_acc(g, q, v, qacc, h=1e-6):
    return (_comvel(g, _integ(q, v, h), v + h * qacc) - _comvel(g, _integ(q, v, -h), v - h * qacc)) / (2 * h)


def _floor_force(sim, g):
    """Sum over floor contacts of the pyramid-edge forces, as a world-frame force on the robot."""
    f, F, n = sim.get("efc_force"), np.zeros(3), 0
    row = sim.geti("nfriction") + sim.geti("nlimit")
    for c in sim.contacts():
        nr, mu = 2 * (c["dim"] - 1) if c["dim"] > 1 else 1, 1.0
        if c["geom1"] == 0:
            fr = c["frame"]
            for k in range(nr):
                F += f[row + k] * (fr[0] + (1 if k % 2 == 0 else -1) * mu * fr[1 + k // 2]) if c["dim"] > 1 else f[row + k] * fr[0]
            n += 1
        row += nr
    return F, n


def test_g1_mass_matrix_and_kinematics_against_numpy():
    g, _ = og.g1_model()
    s = og.G1Sim()
    rng = np.random.default_rng(0)
    for _ in range(3):
        q = g.qpos0.copy()
        q[2] = 1.5
        q[3:7] = rng.normal(size=4)
        q[3:7] /= np.linalg.norm(q[3:7])
        q[7:] = rng.uniform(-0.3, 0.3, 37)
        assert s.set_state(q, rng.normal(size=g.nv)) == 0
        kin = mjcf.forward_kinematics_general(g, q)
        M = mjcf.mass_matrix_general(g, kin)
        qM, Md = s.get("qM"), np.zeros((g.nv, g.nv))
        for i in range(g.nv):
            a, j = g.dof_Madr[i], i
            while j >= 0:
                Md[i, j] = Md[j, i] = qM[a]
                a, j = a + 1, g.dof_parent[j]
        assert np.abs(M - Md).max() < 1e-12
        assert np.abs(kin["xpos"] - s.get("xpos").reshape(-1, 3)).max() < 1e-14
        assert np.abs(kin["geom_xpos"] - s.get("geom_xpos").reshape(-1, 3)).max() < 1e-14


def test_g1_flight_obeys_newton_with_self_contacts_limits_and_friction_loss():
    """No floor contact: whatever the internal constraint forces (mesh-mesh contacts through MPR, joint limits, the 37
    friction-loss rows), the centre of mass must accelerate with g exactly — a check of every Jacobian row's consistency."""
    g, _ = og.g1_model()
    s = og.G1Sim()
    rng = np.random.default_rng(0)
    kinds = set()
    for trial in range(6):
        q = g.qpos0.copy()
        q[2] = 1.5
        q[3:7] = rng.normal(size=4)
        q[3:7] /= np.linalg.norm(q[3:7])
        q[7:] = rng.uniform(-0.6, 0.6, 37) if trial else 0.0
        v = rng.normal(size=g.nv)
        assert s.set_state(q, v) == 0
        assert s.geti("nfriction") == 37
        a = _com_acc(g, q, v, s.get("qacc"))
        assert np.abs(a - [0, 0, -9.81]).max() < 1e-6, a
        for c in s.contacts():
            assert c["geom1"] != 0 and c["dim"] == 3 and c["dist"] < 0
            kinds.add((int(g.geom_type[c["geom1"]]), int(g.geom_type[c["geom2"]])))
        fl = s.get("efc_force")[:37]
        assert (np.abs(fl) <= 0.1 + 1e-12).all()
    assert (MESH, MESH) in kinds


def test_g1_floor_contacts_obey_newton_standing_and_lying():
    """m (a_com - g) = sum of the floor's contact forces, for the walk clip (foot spheres) and the face-down start of the
    getup clip (mesh / cylinder / box against the plane)."""
    g, _ = og.g1_model()
    s = og.G1Sim()
    mtot = g.body_mass.sum()
    types = set()
    for motion, frames in (("walk", (0, 20, 40)), ("getup_facedown", (0, 30, 60))):
        mc = MocapDM(robot="unitree_g1")
        mc.load_mocap(MotionConfig(motion, robot="unitree_g1").mocap_path)
        for fr in frames:
            q, v = np.array(mc.data_config[fr]), np.array(mc.data_vel[fr])
            q[3:7] /= np.linalg.norm(q[3:7])
            for _ in range(60):                 # the retargeted clips float a few mm: lower the pelvis until the floor is felt
                assert s.set_state(q, v) == 0
                if any(c["geom1"] == 0 and c["dist"] < -0.001 for c in s.contacts()):
                    break
                q[2] -= 0.001
            F, n = _floor_force(s, g)
            a = _com_acc(g, q, v, s.get("qacc"))
            assert n > 0 and F[2] > 0
            assert np.abs(mtot * (a - [0, 0, -9.81]) - F).max() < 1e-4 * max(1.0, np.abs(F).max()), (motion, fr, a, F)
            types |= {int(g.geom_type[c["geom2"]]) for c in s.contacts() if c["geom1"] == 0}
    assert SPH in types and MESH in types


def test_g1_friction_loss_rows():
    """Zero gravity, no contacts: a spinning finger joint is opposed by the saturated friction loss (0.1 N m) plus damping;
    a joint moving slowly enough is held inside the box; at rest nothing moves."""
    g, cm = og.g1_model()
    cm2 = og.DmModelG1.from_buffer_copy(cm)
    cm2.gravity[2] = 0.0
    s = og.G1Sim(g, cm2)
    q = g.qpos0.copy()
    q[2] = 1.5
    assert s.set_state(q, np.zeros(g.nv)) == 0
    assert np.abs(s.get("qacc")).max() < 1e-12 and np.abs(s.get("efc_force")).max() < 1e-12
    k = g.jnt_dofadr[g.joint_id("left_knee_joint")]
    v = np.zeros(g.nv)
    v[k] = 5.0
    s.set_state(q, v)
    f = s.get("efc_force")
    assert s.geti("nfriction") == 37 and s.geti("ncon") == 0
    assert abs(f[k - 6] + 0.1) < 1e-9                      # saturated against the motion
    tau = s.get("qfrc_constraint") + s.get("qfrc_passive")
    assert abs(tau[k] - (-0.1 - 0.5 * 5.0)) < 1e-6
    # integrate: the knee velocity decays monotonically and never reverses
    prev = 5.0
    for _ in range(60):
        assert s.step(np.zeros(g.nu)) == 0
        cur = s.get("qvel")[k]
        assert -1e-9 <= cur <= prev + 1e-9
        prev = cur
    assert prev < 1.0


# ------------------------------------------------------------------------------------------ task layer
@pytest.fixture(scope="module")
def walk():
    mc = MocapDM(robot="unitree_g1")
    mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
    return mc, og.G1Clip(*mc.tables())


def _rpy(q):
    w, x, y, z = q
    return (math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), math.asin(max(-1, min(1, 2 * (w * y - z * x)))),
            math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))


def _reward_numpy(g, sim, mc, idx):
    """src/deepmimic_env.py:193-256 with the unitree_g1 branch, on the oracle's state."""
    rc = RobotConfig("unitree_g1")
    qi, vi = np.array(og.REW_QPOS), np.array(og.REW_QVEL)
    qpos, qvel = sim.get("qpos"), sim.get("qvel")
    tq, tv = np.array(mc.get_qpos(idx)), np.array(mc.get_qvel(idx))
    err = np.abs(qpos[qi] - tq[qi]).sum() + abs(_rpy(qpos[3:7])[1] - _rpy(tq[3:7])[1])
    r_cfg, r_vel = math.exp(-err), math.exp(-0.1 * np.abs(tv[vi] - qvel[vi]).sum())
    gx = sim.get("geom_xpos").reshape(-1, 3)
    ee = sum(np.linalg.norm(gx[g.geom_id(n)] - mc.get_geom_xpos(idx)[g.geom_id(n)]) ** 2 for n in rc.endeffector_geom_names)
    mass = g.body_mass[:, None]
    com_err = np.linalg.norm((mc.get_body_xpos(idx) * mass).sum(0) / mass.sum()
                             - (sim.get("xpos").reshape(-1, 3) * mass).sum(0) / mass.sum()) ** 2
    tol = (g.jnt_range[1:] * 0.99)[qi - 7]
    jp = qpos[7:][qi - 7]
    qlim = ((jp <= tol[:, 0]).sum() + (jp >= tol[:, 1]).sum()) / len(jp)
    terms = np.array([r_cfg, r_vel, math.exp(-40 * ee), math.exp(-10 * com_err), qlim])
    return 0.75 * terms[0] + 0.1 * terms[1] + 0.15 * terms[2] - 0.1 * terms[4], terms


def test_g1_env_observation_reward_and_action_padding(walk):
    mc, clip = walk
    g, _ = og.g1_model()
    s = og.G1Sim()
    obs, err = s.env_reset(clip, 7)
    assert err == 0 and obs.shape == (85,)
    q, v = s.get("qpos"), s.get("qvel")
    assert np.allclose(obs[:37], q[7:]) and np.allclose(obs[37:74], 0.1 * v[6:])
    r, p, _ = _rpy(s.get("xquat").reshape(-1, 4)[1])
    assert np.allclose(obs[74:76], [0.1 * r, 0.1 * p])
    assert obs[82] == 0 and obs[83] == 0          # G1 "foot" geoms are visual: the contact flags never fire
    assert abs(obs[84] - 7 / clip.L) < 1e-15
    rng = np.random.default_rng(3)
    for k in range(12):
        act = rng.uniform(-1, 1, 23)
        idx = s.env.idx_curr
        obs, rew, done, terms, reason = s.env_step(clip, act)
        ctrl = s.get("ctrl")
        assert np.allclose(ctrl[:23], 20.0 * act) and (ctrl[23:] == 0).all()      # :348-351
        rew_np, terms_np = _reward_numpy(g, s, mc, idx)
        assert abs(rew - rew_np) < 1e-12 and np.abs(terms - terms_np).max() < 1e-12
        assert s.env.idx_curr == (idx + 1) % clip.L
        if done:
            break
    # teacher-forced onto the clip the imitation terms are all exactly 1
    obs, _ = s.env_reset(clip, 30)
    fq, fv = np.array(mc.get_qpos(30)), np.array(mc.get_qvel(30))
    obs, rew, done, terms, reason = s.env_step(clip, np.zeros(23), force_state=(fq, fv))
    assert np.allclose(terms[:4], 1.0, atol=1e-6) and abs(rew - (1.0 - 0.1 * terms[4])) < 1e-6   # (set_state renormalises the root quaternion)


def test_g1_termination_rules(walk):
    mc, clip = walk
    s = og.G1Sim()
    # low_z is 0.4 for this robot (src/config.py:22): a pelvis at 0.55 m is alive, at 0.3 m it is not
    for z, dead in ((0.55, False), (0.30, True)):
        s.env_reset(clip, 0)
        fq, fv = np.array(mc.get_qpos(0)), np.zeros(43)
        fq[2] = z
        _, _, done, _, reason = s.env_step(clip, np.zeros(23), force_state=(fq, fv))
        assert done == dead and reason == (1 if dead else 2)
    # the G1 "run" rule (:426-433): roll or pitch more than 60 degrees off the clip's ends the episode
    run = MocapDM(robot="unitree_g1")
    run.load_mocap(MotionConfig("run", robot="unitree_g1").mocap_path)
    for rule, expect in ((True, True), (False, False)):
        rclip = og.G1Clip(*run.tables(), run_rule=rule)
        s.env_reset(rclip, 0)
        fq, fv = np.array(run.get_qpos(0)), np.zeros(43)
        ang = math.radians(70.0)
        fq[3:7] = quat_mul(fq[3:7] / np.linalg.norm(fq[3:7]), [math.cos(ang / 2), math.sin(ang / 2), 0, 0])
        fq[2] = 1.0
        _, _, done, _, reason = s.env_step(rclip, np.zeros(23), force_state=(fq, fv))
        assert done == expect and (reason == 8) == expect


def test_g1_passive_rollout_regime(walk):
    """Reference-state init on the walk clip, zero torques: the robot keeps its feet on the floor for a few steps, then
    sinks below low_z (0.4 m) within a second; no solver blow-up on the way (contacts via spheres, meshes and MPR)."""
    mc, clip = walk
    s = og.G1Sim()
    obs, err = s.env_reset(clip, 0)
    assert err == 0 and s.geti("ncon") >= 1
    steps, reason = 0, 0
    for steps in range(1, 80):
        obs, rew, done, terms, reason = s.env_step(clip, np.zeros(23))
        assert np.isfinite(obs).all() and reason != 5 and s.geti("overflow_con") == 0 and s.geti("overflow_row") == 0
        if done:
            break
    assert 8 <= steps <= 60 and reason == 1


# ------------------------------------------------------------------------------------------ DPCombinedEnv on the G1
@pytest.fixture(scope="module")
def comb_clips():
    clips, mocaps = [], []
    for m in ("walk", "run", "getup_facedown_towalk"):
        mc = MocapDM(robot="unitree_g1")
        mc.load_mocap(MotionConfig(m, robot="unitree_g1").mocap_path)
        mocaps.append(mc)
        clips.append(og.G1Clip(*mc.tables()))
    return clips, mocaps


def test_g1_combined_env_state_machine_on_the_oracle(comb_clips):
    """DPCombinedEnv as the reference runs it (src/combined_env.py on the G1): obs 98 with the extra-contact block, getup runs
    out of time and hands over to RUN (the `== PAWalk()` identity quirk, :396), a robot that falls without amnesty ends the
    episode and enters to_getup; teacher-forced onto the clip the imitation terms are 1 and walk earns the velocity reward."""
    clips, mocaps = comb_clips
    s = og.G1CombSim(clips)
    obs, err = s.comb_reset(2, 0)
    assert err == 0 and obs.shape == (98,)
    assert list(obs[93:98]) == [1, 0, 0, 0, 1] and abs(obs[91] - 1) < 1e-6 and abs(obs[90]) < 1e-12   # one-hot PAWalk, getup flag
    L = clips[2].L
    seen_contact = False
    for t in range(L + 5):
        obs, r, d, terms, reason = s.comb_step(np.zeros(23))
        seen_contact |= obs[82:90].any()
        assert set(np.unique(obs[82:90])) <= {0.0, 1.0}
        if d:
            break
    assert seen_contact
    assert t == L - 1 and d and reason == 7          # out of time at n_steps = L - 1 -> run -> fallen (lying) without amnesty
    assert s.cenv.motion == 3 and s.cenv.n_steps == 1
    # teacher-forced walk: all imitation terms 1, task reward 1 (root velocity of the clip), reward 0.7 + 0.3
    s.comb_reset(0, 200)
    fr = 200 % clips[0].L
    fq, fv = np.array(mocaps[0].get_qpos(fr)), np.array(mocaps[0].get_qvel(fr))
    obs, r, d, terms, reason = s.comb_step(np.zeros(23), force_state=(fq, fv))
    assert not d and np.allclose(terms[:4], 1, atol=1e-6) and abs(terms[6] - 1) < 1e-9 and abs(r - (0.7 * (1 - 0.1 * terms[4]) + 0.3)) < 1e-6
    assert s.cenv.motion == 0 and s.cenv.n_steps == 201
    # to_getup: success within 15 degrees of frame 1 of the getup clip hands over to getup
    s.comb_reset(3, 5)
    fq, fv = np.array(mocaps[2].get_qpos(1)), np.array(mocaps[2].get_qvel(1))
    obs, r, d, terms, reason = s.comb_step(np.zeros(23), force_state=(fq, fv))
    assert obs[96] == 1 and obs[97] == 0 and terms[5] == 0 and abs(terms[6] - 1 / 3) < 1e-6    # imitation 0, exp(0) / 3
    assert s.cenv.motion == 2 and s.cenv.n_steps == 1
