#!/usr/bin/env python3
"""Marginal cost of the solver: time the step kernel with the Newton iteration cap at 0, 1, 2, 3 and unlimited."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions

N = 4096
cfg = make_config("flamingo_light_v1", num_envs=N, seed=1234)
for cap in (50, 0, 1, 2, 3, 4, 50):
    env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
    env.engine.set_param("max_newton", np.array([cap], dtype=np.float32))
    acts = synthetic_actions(N, 0, 160, 4, env.device)
    env.reset()
    for t in range(60):
        env.step(acts[t])
    env.engine.set_timing(True)
    for t in range(60, 160):
        env.step(acts[t])
    torch.cuda.synchronize()
    ms, n = env.engine.kernel_time()
    st = env.solver_stats()
    nsub = (st["step_count"] - N) * 4
    print(f"max_newton={cap:3d}: kernel {ms*1e3:8.1f} us  newton/substep {st['newton_iters']/nsub:.2f} ls/substep {st['ls_evals']/nsub:.2f} fact/substep {st['factorisations']/nsub:.2f} rows {st['rows']/nsub:.1f} nan_resets {st['nan_resets']}", flush=True)
    env.close()
