#!/usr/bin/env python3
"""The driver's short bench run (20 timed steps after 5 warm-up) taken apart: kernel-timing events on / off, fix-up launches on / off."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from bench import synthetic_actions, workload_config
N, W, K = 4096, 5, 20
cfg = workload_config("light_flat", N)
def run(timing, fixup, stride=1, reps=9):
    env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, ranges=4, deferred_join=True)
    if not fixup:
        env.engine.set_param("fixup", np.array([0.0], dtype=np.float32))
    if stride > 1:
        env.engine.set_param("timing_stride", np.array([float(stride)], dtype=np.float32))
    acts = synthetic_actions(N, 0, W + K, env.action_dim, env.device)
    out = []
    for r in range(reps):
        env.reset()
        for t in range(W):
            env.step(acts[t])
        env.join(); torch.cuda.synchronize()
        env.engine.set_timing(timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            env.step(acts[t])
        env.join(); torch.cuda.synchronize()
        out.append(N * K / (time.perf_counter() - t0) / 1e6)
        env.engine.set_timing(False)
    env.close()
    first = out[0]
    out = np.array(out[1:])
    print(f"(first repetition, right after start-up: {first:6.2f} M)  timing {int(timing)} stride {stride} fixup {int(fixup)}: median {np.median(out):6.2f} M  min {out.min():6.2f}  max {out.max():6.2f}", flush=True)
run(True, True, stride=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
run(True, True, stride=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
