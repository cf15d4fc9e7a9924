#!/usr/bin/env python3
"""Does splitting the fleet into independently stepped shards (one HIP stream each) hide the launch tail?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import make_config
from bench import synthetic_actions

N, K, W = 4096, 1000, 100
for shards in (1, 2, 4, 8):
    n = N // shards
    cfg = make_config("flamingo_light_v1", num_envs=n, seed=1234)
    cm = compile_model(cfg)
    envs = [BatchedEnv(cfg, num_envs=n, seed=1234, auto_reset=True, env_id0=i * n, gain_noise=0.1, compiled=cm) for i in range(shards)]
    streams = [torch.cuda.Stream() for _ in range(shards)]
    acts = [synthetic_actions(n, i * n, K + W, 4, envs[i].device) for i in range(shards)]
    torch.cuda.synchronize()
    for i, e in enumerate(envs):
        with torch.cuda.stream(streams[i]):
            e.reset()
            for t in range(W):
                e.step(acts[i][t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        for i, e in enumerate(envs):
            with torch.cuda.stream(streams[i]):
                e.step(acts[i][t])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"shards {shards}: {N*K/dt/1e6:.2f} M env-steps/s  ({dt/K*1e3:.3f} ms per fleet step)", flush=True)
    for e in envs:
        e.close()
