#!/usr/bin/env python3
"""Solver tolerance sweep: step-kernel time in the bench regime and one-step replay error against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.model import get_field
from oracle.oracle import Oracle
from bench import synthetic_actions

# replay set
cfgp = make_config("flamingo_light_v1", random=PARITY_RANDOM)
cm = compile_model(cfgp)
o = Oracle(cm)
o.reset(np.array(get_field(cm.blob, "init_qpos")[:19]))
R = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[])
for t in range(400):
    a = 0.25 * np.sin(2 * np.pi * 0.5 * 0.02 * t + np.array([0.0, 1.0, 2.0, 3.0]))
    R["qpos"].append(o.qpos.copy()); R["qvel"].append(o.qvel.copy()); R["warm"].append(o.qacc_warmstart.copy()); R["act"].append(a)
    o.control_step(a)
    R["qpos1"].append(o.qpos.copy()); R["qvel1"].append(o.qvel.copy())
R = {k: np.array(v) for k, v in R.items()}

N = 4096
cfg = make_config("flamingo_light_v1", num_envs=N, seed=1234)
for tol, mls in ((1e-6, 24), (1e-6, 12), (1e-6, 8), (1e-6, 6), (1e-6, 4), (1e-6, 3)):
    env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
    env.engine.set_param("solver_tolerance", np.array([tol], dtype=np.float32))
    env.engine.set_param("max_ls", np.array([mls], dtype=np.float32))
    acts = synthetic_actions(N, 0, 300, 4, env.device)
    env.reset()
    for t in range(100):
        env.step(acts[t])
    env.engine.set_timing(True)
    for t in range(100, 300):
        env.step(acts[t])
    torch.cuda.synchronize()
    ms, n = env.engine.kernel_time()
    st = env.solver_stats(); nsub = (st["step_count"] - N) * 4
    env.close()
    envp = BatchedEnv(cfgp, num_envs=400, auto_reset=False, compiled=cm)
    envp.engine.set_param("solver_tolerance", np.array([tol], dtype=np.float32))
    envp.engine.set_param("max_ls", np.array([mls], dtype=np.float32))
    envp.reset(); envp.set_state(R["qpos"], R["qvel"], R["warm"])
    envp.step(torch.tensor(R["act"], dtype=torch.float32, device=envp.device))
    d = envp.get_data()
    ev = np.abs(d.qvel.cpu().numpy() - R["qvel1"]).max(axis=1); ep = np.abs(d.qpos.cpu().numpy() - R["qpos1"]).max(axis=1)
    envp.close()
    print(f"tol {tol:7.0e} max_ls {mls:2d}: kernel {ms*1e3:7.1f} us  newton {st['newton_iters']/nsub:.2f} ls {st['ls_evals']/nsub:.2f} | replay dqvel max {ev.max():.2e} median {np.median(ev):.2e} dqpos max {ep.max():.2e}", flush=True)
