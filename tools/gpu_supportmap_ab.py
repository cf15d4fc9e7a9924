#!/usr/bin/env python3
"""Support maps on / off: fleet throughput of the mesh-geom workloads (plain env.step loop, engine ranges)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from cosim_amd.batched_env import BatchedEnv
from bench import synthetic_actions, workload_config, WORKLOADS

K = 100
for wl in sys.argv[1:] or ["w4_rocky", "p_v3_flat", "humanoid_stairs", "humanoid_flat"]:
    N = WORKLOADS[wl][3]
    cfg = workload_config(wl, N)
    for use_map in (1.0, 0.0):
        env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, ranges=4, deferred_join=True)
        env.engine.set_param("support_map", np.array([use_map], dtype=np.float32))
        acts = synthetic_actions(N, 0, 50 + K, env.action_dim, env.device)
        if wl == "humanoid_stairs":
            env.receive_user_command(np.random.default_rng(0).uniform(-3, 3, size=(N, 2)).astype(np.float32))
        env.reset()
        for t in range(50):
            env.step(acts[t])
        env.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(50, 50 + K):
            env.step(acts[t])
        env.join(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = env.solver_stats()
        print(f"{wl:18s} support_map={int(use_map)}: {N*K/(t1-t0)/1e6:7.3f} M env-steps/s   rows {st['rows']} newton {st['newton_iters']}", flush=True)
        env.close()
