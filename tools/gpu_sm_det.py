#!/usr/bin/env python3
"""humanoid on stairs: is the fleet reproducible run to run, with and without the support maps?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
n, K = 128, 60
cfg = make_config("humanoid_p_v0", terrain="stairs_up_hard", num_envs=n, seed=33)
def run(use_map, split):
    env = BatchedEnv(cfg, num_envs=n, seed=33, auto_reset=True, gain_noise=0.1)
    env.engine.set_param("support_map", np.array([use_map], dtype=np.float32))
    if not split:
        env.engine.set_param("split", np.array([0.0], dtype=np.float32))
    acts = (0.5 * torch.randn((K, n, env.action_dim), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(8))).clamp_(-1, 1)
    env.reset()
    first = None
    S = []
    for k in range(K):
        s, _, _, _ = env.step(acts[k]); S.append(s.clone())
    st = env.solver_stats(); env.close()
    return torch.stack(S), st["rows"]
for split in (True, False):
    a, ra = run(1.0, split); b, rb = run(1.0, split); c, rc = run(0.0, split); d, rd = run(0.0, split)
    def first_diff(x, y):
        ne = (x != y).any(dim=2)
        if not ne.any(): return None
        k = int(ne.any(dim=1).nonzero()[0]); return k, int(ne[k].sum())
    print("split", split, "rows", ra, rb, rc, rd, " map/map", first_diff(a, b), " scan/scan", first_diff(c, d), " map/scan", first_diff(a, c), flush=True)
