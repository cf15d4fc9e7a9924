#!/usr/bin/env python3
"""Hulls with few prisms under them (coarse terrain): staged lane-parallel walk (support maps) against the wave-cooperative walk."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from cosim_amd.batched_env import BatchedEnv
from bench import synthetic_actions, workload_config, WORKLOADS
K = 100
res = {}
for wl in sys.argv[1:] or ["w4_rocky"]:
    N = WORKLOADS[wl][3]
    cfg = workload_config(wl, N)
    for coop in (0.0, 1.0):
        env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, ranges=4, deferred_join=True)
        env.engine.set_param("coop_walk", np.array([coop], dtype=np.float32))
        acts = synthetic_actions(N, 0, 50 + K, env.action_dim, env.device)
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
        res[coop] = env.state.clone()
        print(f"{wl:12s} coop_walk={int(coop)}: {N*K/(t1-t0)/1e6:7.3f} M env-steps/s   rows {st['rows']} newton {st['newton_iters']} dropped {st['dropped_contacts']}", flush=True)
        env.close()
    print("   same bits:", bool(torch.equal(res[0.0], res[1.0])), " max |diff|", float((res[0.0] - res[1.0]).abs().max()))
