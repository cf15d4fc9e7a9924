#!/usr/bin/env python3
"""Where does the wall time per step go beyond the kernel?  Loop variants: full env.step with/without event timing, raw C ABI call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions

N, K = 4096, 600
cfg = make_config("flamingo_light_v1", num_envs=N, seed=1234)
env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
acts = synthetic_actions(N, 0, K + 100, 4, env.device)
env.reset()
for t in range(100):
    env.step(acts[t])
torch.cuda.synchronize()

def run(label, timing, raw):
    env.engine.set_timing(timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if raw:
        e = env.engine
        st = env._stream()
        ptrs = (env._cmd_ptr(), env.state.data_ptr(), env.terminated.data_ptr(), env.truncated.data_ptr(), env.info_buf.data_ptr())
        for t in range(100, 100 + K):
            e.step(acts[t].data_ptr(), *ptrs, st)
    else:
        for t in range(100, 100 + K):
            env.step(acts[t])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms, n = env.engine.kernel_time() if timing else (0.0, 0)
    print(f"{label:40s} host-loop {1e6*(t1-t0)/K:7.1f} us/step  wall {1e6*(t2-t0)/K:7.1f} us/step  kernel {ms*1e3:7.1f} us", flush=True)

run("env.step, event timing on", True, False)
run("env.step, timing off", False, False)
run("raw C ABI, timing off", False, True)
run("raw C ABI, timing on", True, True)
