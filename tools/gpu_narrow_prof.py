#!/usr/bin/env python3
"""Where the narrowphase kernel of the split pipeline (humanoid_p_v0 on stairs) spends its waves: lifetimes (mean, max), the walk's
phases, items.  Diagnostic build (cosim_set_param narrow_occupancy 0).   python tools/gpu_narrow_prof.py [waves_per_env] [settle]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from bench import synthetic_actions, workload_config
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 22
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N = 1024
cfg = workload_config("humanoid_stairs", N)
env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
env.engine.set_param("narrow_waves", np.array([float(nw)]))
acts = synthetic_actions(N, 0, settle + 40, env.action_dim, env.device)
from cosim_amd import rng as crng
gids = np.arange(N, dtype=np.uint64)[:, None]
env.receive_user_command((6.0 * crng.uniform(1234, gids, 0, 6, np.arange(2)[None, :]) - 3.0).astype(np.float32))
env.reset()
for t in range(settle):
    env.step(acts[t])
torch.cuda.synchronize()
env.engine.set_param("narrow_occupancy", np.array([0.0]))
env.engine.debug_counters(clear=True)
K = 20
for t in range(settle, settle + K):
    env.step(acts[t])
c = env.engine.debug_counters().astype(np.float64)
launches = K * 4
print(f"narrowphase kernel, {nw} waves per env, {N} envs, after {settle} steps: {c[2]/launches:.0f} waves per launch")
print(f"  wave lifetime: mean {c[0]/c[2]:.0f} cycles, max {c[1]:.0f} cycles ({c[1]/2.4e3:.0f} us at 2.4 GHz); sum over waves per launch {c[0]/launches/1e6:.1f} M cycles "
      f"= {c[0]/launches/1024/2.4e3:.0f} us if spread evenly over 1024 SIMDs")
print(f"  work items per launch {c[3]/launches:.0f}; waves living > 400 k cycles: {c[4]/launches:.1f} per launch, {c[5]/max(c[4],1):.0f} items each, "
      f"sub-grid+height {c[6]/max(c[4],1):.0f} cycles, probe+full {c[7]/max(c[4],1):.0f} cycles each")
names = ["sub-grids + height passes", "probe passes", "full-MPR batches", "slowest-lane MPR iterations (sum)", "probe batches", "cycles in MPR", "full batches", "set-up before MPR"]
for i, n in enumerate(names):
    print(f"  {n:36s} {c[8+i]/launches:14.0f} per launch   {c[8+i]/c[2]:10.0f} per wave")
