#!/usr/bin/env python3
"""Device support routines, map against scan, on every mesh hull of a robot."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from cosim_amd.model import get_field
for robot in sys.argv[1:] or ["humanoid_p_v0", "w4_p_v2"]:
    env = BatchedEnv(make_config(robot, num_envs=2, seed=1), num_envs=2, seed=1)
    b = env.cm.blob
    gnum = np.array(get_field(b, "geom_hullnum")[:b.ngeom]); gadr = np.array(get_field(b, "geom_hulladr")[:b.ngeom])
    rng = np.random.default_rng(0)
    for g in range(b.ngeom):
        if gnum[g] < 32: continue
        V = env.cm.hull_vert[gadr[g]:gadr[g] + gnum[g]]
        D = rng.normal(size=(20000, 3))
        D[:2000, :2] *= 1e-4      # near -z / +z
        D[2000:4000, 1:] *= 1e-4
        D = (D / np.linalg.norm(D, axis=1)[:, None]).astype(np.float32)
        outs = []
        for use_map in (1, 0):
            o = np.zeros((len(D), 6), dtype=np.float32)
            env.engine._check(env.engine.L.cosim_debug_support(env.engine.h, g, D.ctypes.data, len(D), o.ctypes.data, use_map))
            outs.append(o)
        lane_ne = (outs[0][:, :3] != outs[1][:, :3]).any(1); coop_ne = (outs[0][:, 3:] != outs[1][:, 3:]).any(1)
        lc = (outs[1][:, :3] != outs[1][:, 3:]).any(1)
        print(robot, "geom", g, "verts", gnum[g], "lane-parallel map != scan:", int(lane_ne.sum()), " cooperative map != scan:", int(coop_ne.sum()), " scan lane != scan coop:", int(lc.sum()))
        for i in np.nonzero(lane_ne)[0][:3]:
            print("   dir", D[i], "map", outs[0][i, :3], "scan", outs[1][i, :3], "dots", D[i] @ outs[0][i, :3], D[i] @ outs[1][i, :3])
    env.close()
