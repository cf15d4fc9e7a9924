#!/usr/bin/env python3
"""cosim_rollout (K steps per launch) against the cosim_step loop, same fleet, same action table."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions, workload_config, WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "light_flat"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N = WORKLOADS[wl][3]
cfg = workload_config(wl, N)
for ranges in (1, 2, 4):
    env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, ranges=ranges, deferred_join=ranges > 1)
    acts = synthetic_actions(N, 0, 50 + 2 * K, env.action_dim, env.device)
    if env.engine.query("rollout") != 1:
        print(f"{wl}: no rollout kernel for this model / terrain (flat flamingo_light_v1 / flamingo_p_v3 only)")
        env.close()
        break
    env.reset()
    for t in range(50):
        env.step(acts[t])
    env.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(50, 50 + K):
        env.step(acts[t])
    env.join(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = env.rollout(acts[50 + K:50 + 2 * K])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for chunk in (10, 50):
        tc0 = time.perf_counter()
        for c in range(0, K, chunk):
            env.rollout(acts[50 + c:50 + c + chunk], info=False)
        torch.cuda.synchronize()
        tc1 = time.perf_counter()
        print(f"   chunks of {chunk}: {N*K/(tc1-tc0)/1e6:7.3f} M", flush=True)
    st = env.solver_stats()
    print(f"{wl} ranges={ranges}: step loop {N*K/(t1-t0)/1e6:7.3f} M   rollout({K}) {N*K/(t2-t1)/1e6:7.3f} M env-steps/s   fixups {st['fixup_steps']} dropped {st.get('dropped_contacts')}", flush=True)
    env.close()
