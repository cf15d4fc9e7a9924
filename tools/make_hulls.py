#!/usr/bin/env python3
"""Reduce the reference's collision STL meshes to convex-hull vertex lists.

Run in the build container only (it reads /root/reference/envs/<robot>/assets/mesh/*.STL,
which does not travel).  Output: cosim_amd/assets/<robot>/hulls.npz holding, per mesh
name used by a collision geom of the MJCF, the convex-hull vertices (float32, mesh
coordinates) and the hull's vertex-adjacency graph in CSR form (neighbours sorted by
vertex index).  MuJoCo builds the same hull with qhull when it compiles a mesh geom that
can collide [upstream: user/user_mesh.cc MakeGraph]; only the hull takes part in
plane/hfield-vs-mesh collision (engine_collision_convex.c mjc_PlaneConvex), so the
48 MB of raw triangles are not needed at run time.

STLs absent from the reference checkout get a documented primitive proxy
(a box given as 8 hull vertices); the proxy table is below.
"""
import glob
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial import ConvexHull

REF = "/root/reference/envs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cosim_amd", "assets")

# proxy boxes for meshes missing from the checkout: (centre xyz, half-extent xyz) in mesh coordinates
PROXY = {
    "flamingo_light_v1": {
        "base_link_fixed.STL": ((-0.02, 0.0, -0.04), (0.10, 0.09, 0.06)),
    },
    "flamingo_p_v3": {
        "base_link_fixed.STL": ((0.0, 0.0, 0.0), (0.10, 0.10, 0.06)),
        "left_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "right_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "left_shoulder_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "right_shoulder_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
    },
    "w4_p_v2": {
        "base_link_fixed.STL": ((0.0, 0.0, 0.0), (0.25, 0.12, 0.06)),
        "FL_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "FR_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "RL_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "RR_hip_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "FL_shoulder_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
        "RL_shoulder_link.STL": ((0.0, 0.0, 0.0), (0.04, 0.04, 0.04)),
    },
    "humanoid_p_v0": {},
}


def read_stl(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    if len(b) != 84 + 50 * n:
        raise ValueError(f"{path}: not a binary STL")
    rec = np.frombuffer(b[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return rec["v"].reshape(-1, 3)


def hull_of(points):
    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    h = ConvexHull(pts)
    idx = np.sort(h.vertices)
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[idx] = np.arange(len(idx))
    nbrs = [set() for _ in idx]
    for tri in h.simplices:
        t = remap[tri]
        for a in range(3):
            for b in range(3):
                if a != b:
                    nbrs[t[a]].add(int(t[b]))
    adr = np.zeros(len(idx) + 1, dtype=np.int32)
    flat = []
    for i, s in enumerate(nbrs):
        flat.extend(sorted(s))
        adr[i + 1] = len(flat)
    return pts[idx].astype(np.float32), adr, np.asarray(flat, dtype=np.int32)


def box_points(c, h):
    c = np.asarray(c)
    h = np.asarray(h)
    return np.array([[c[0] + sx * h[0], c[1] + sy * h[1], c[2] + sz * h[2]]
                     for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def collision_meshes(robot):
    """mesh asset names referenced by geoms that can collide (contype|conaffinity != 0)."""
    from cosim_amd.mjcf import parse_mjcf  # local import: tools run from the repo root
    spec = parse_mjcf(os.path.join(OUT, robot, f"{robot}.xml"))
    names = []
    for g in spec["geoms"]:
        if g["type"] == "mesh" and (g["contype"] or g["conaffinity"]) and g["mesh"] not in names:
            names.append(g["mesh"])
    return names, spec["mesh_files"]


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    for robot in ("flamingo_light_v1", "flamingo_p_v3", "w4_p_v2", "humanoid_p_v0"):
        names, files = collision_meshes(robot)
        out = {}
        for name in names:
            path = os.path.normpath(os.path.join(REF, robot, "assets", "xml", files[name]))
            if os.path.isfile(path):
                v, adr, nbr = hull_of(read_stl(path))
                src = "stl"
            elif os.path.basename(path) in PROXY[robot]:
                v, adr, nbr = hull_of(box_points(*PROXY[robot][os.path.basename(path)]))
                src = "proxy-box"
            else:
                raise FileNotFoundError(f"{robot}: {name} missing and no proxy defined")
            out[f"{name}/vert"] = v
            out[f"{name}/adr"] = adr
            out[f"{name}/nbr"] = nbr
            out[f"{name}/src"] = np.array(src)
            print(f"{robot:20s} {name:28s} {src:10s} hull vertices {len(v):4d} edges {len(nbr)//2}")
        np.savez_compressed(os.path.join(OUT, robot, "hulls.npz"), **out)


if __name__ == "__main__":
    main()
