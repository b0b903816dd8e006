"""Convex hulls of the Unitree G1 collision meshes -> deepmimic_mujoco_amd/assets/unitree_g1_hulls.npz.

MuJoCo collides a mesh geom as the convex hull of its vertices (qhull at compile time) [EXT]; the hull vertices are all
the G1 model compiler (deepmimic_mujoco_amd/mjcf.py) needs of the 17 MB of binary STL files next to the reference's
deepmimic_unitree_g1.xml (meshdir "assets").  This script reads those DATA files (binary STL: 80-byte header, uint32
triangle count, 50-byte records) and writes, per mesh name, the hull vertices (float32, mesh frame) and the hull's
triangle indices.  Run here, where /root/reference exists; the npz travels with the repo.
"""
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

REF_ASSET = "/root/reference/src/mujoco/humanoid_deepmimic/envs/asset"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "deepmimic_mujoco_amd", "assets", "unitree_g1_hulls.npz")


def read_stl(path):
    raw = open(path, "rb").read()
    n = struct.unpack_from("<I", raw, 80)[0]
    assert len(raw) == 84 + 50 * n, path
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return rec["v"].reshape(-1, 3).astype(np.float64)


def main():
    xml = os.path.join(REF_ASSET, "deepmimic_unitree_g1.xml")
    root = ET.parse(xml).getroot()
    meshdir = root.find("compiler").get("meshdir", "")
    used = {g.get("mesh") for g in root.iter("geom") if g.get("mesh") and g.get("class") == "collision"}
    out = {}
    tot = 0
    for me in root.find("asset").findall("mesh"):
        f = me.get("file")
        name = me.get("name", os.path.splitext(f)[0])
        if name not in used:
            continue
        v = np.unique(read_stl(os.path.join(REF_ASSET, meshdir, f)), axis=0)
        h = ConvexHull(v)
        idx = np.sort(h.vertices)
        remap = -np.ones(len(v), np.int64)
        remap[idx] = np.arange(len(idx))
        out[name] = v[idx].astype(np.float32)
        out[name + "__faces"] = remap[h.simplices].astype(np.int32)
        tot += len(idx)
        print("%-28s %6d verts -> hull %4d verts, %4d faces" % (name, len(v), len(idx), len(h.simplices)))
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes;", tot, "hull vertices in", len(out) // 2, "meshes")


if __name__ == "__main__":
    sys.exit(main())
