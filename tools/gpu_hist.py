#!/usr/bin/env python3
"""Distribution over envs of Newton iterations / line-search evaluations per control step (tail of the launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions
N = 4096
cfg = make_config("flamingo_light_v1", num_envs=N, seed=1234)
env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
acts = synthetic_actions(N, 0, 400, 4, env.device)
env.reset()
def meta():
    buf = torch.zeros((N, 8), dtype=torch.float32, device=env.device)
    env.engine.get("meta", buf.data_ptr(), env._stream()); torch.cuda.synchronize()
    return buf.view(torch.int32).cpu().numpy().astype(np.int64)
for t in range(300):
    env.step(acts[t])
hn, hl = [], []
for t in range(300, 330):
    m0 = meta(); env.step(acts[t]); m1 = meta()
    hn.append(m1[:, 5] - m0[:, 5]); hl.append(m1[:, 6] - m0[:, 6])
hn, hl = np.array(hn), np.array(hl)
print("newton iters per control step: mean %.1f  p50 %d p90 %d p99 %d max-per-launch mean %.1f (max %d)" % (hn.mean(), np.median(hn), np.quantile(hn, .9), np.quantile(hn, .99), hn.max(axis=1).mean(), hn.max()))
print("ls evals per control step:     mean %.1f  p50 %d p90 %d p99 %d max-per-launch mean %.1f (max %d)" % (hl.mean(), np.median(hl), np.quantile(hl, .9), np.quantile(hl, .99), hl.max(axis=1).mean(), hl.max()))
print("hist newton", np.bincount(hn.ravel().astype(int))[:60])
