#!/usr/bin/env python3
"""Where does the wall time per step go beyond the kernel?  Host-loop time (enqueue only) against wall time per control step for:
plain env.step with engine-owned ranges (deferred join), the same through the raw C ABI, and caller-built streams + env.step_range."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions

N, K = 4096, 600
cfg = make_config("flamingo_light_v1", num_envs=N, seed=1234)
acts = None


def run(label, ranges, timing, mode):
    global acts
    env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, ranges=ranges if mode != "caller" else 1,
                     deferred_join=ranges > 1 and mode != "caller")
    if acts is None:
        acts = synthetic_actions(N, 0, K + 100, 4, env.device)
    env.reset()
    S = ranges
    streams = [torch.cuda.Stream(device=env.device) for _ in range(S)] if mode == "caller" else None
    rl = [(i * (N // S), N // S) for i in range(S)]

    def step(t):
        if mode == "caller":
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    env.step_range(rl[i][0], rl[i][1], acts[t])
        elif mode == "raw":
            env.engine.step(acts[t].data_ptr(), *ptrs, st)
        else:
            env.step(acts[t])
    st = env._stream()
    ptrs = (env._cmd_ptr(), env.state.data_ptr(), env.terminated.data_ptr(), env.truncated.data_ptr(), env.info_buf.data_ptr())
    for t in range(100):
        step(t)
    torch.cuda.synchronize()
    env.engine.set_timing(timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(100, 100 + K):
        step(t)
    t1 = time.perf_counter()
    env.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ms, n = env.engine.kernel_time() if timing else (0.0, 0)
    print(f"{label:52s} host-loop {1e6*(t1-t0)/K:7.1f} us/step  wall {1e6*(t2-t0)/K:7.1f} us/step  {N*K/(t2-t0)/1e6:6.2f} M  kernel {ms*1e3:6.1f} us", flush=True)
    env.close()


sel = sys.argv[1:] or ["all"]
run("env.step, 4 engine ranges, timing on", 4, True, "env")
run("env.step, 4 engine ranges, timing off", 4, False, "env")
run("raw cosim_step, 4 engine ranges, timing off", 4, False, "raw")
run("caller streams + step_range x 4, timing on", 4, True, "caller")
run("caller streams + step_range x 4, timing off", 4, False, "caller")
run("env.step, 1 launch, timing off", 1, False, "env")
run("env.step, 2 engine ranges, timing off", 2, False, "env")
run("env.step, 3 engine ranges, timing off", 3, False, "env")
