#!/usr/bin/env python3
"""Per-phase shader-clock profile of the step kernel (diagnostic build with s_memtime stamps; shares, not run times)."""
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
acts = synthetic_actions(N, 0, 340, 4, env.device)
env.reset()
for t in range(300):
    env.step(acts[t])
torch.cuda.synchronize()
acc = np.zeros(16)
K = 20
for t in range(300, 300 + K):
    a = acts[t].contiguous()
    acc += env.engine.profile_step(a.data_ptr(), env._cmd_ptr(), env.state.data_ptr(), env.terminated.data_ptr(), env.truncated.data_ptr())
acc /= K
names = ["prologue", "kinematics", "comPos+cdof", "crb (M)", "comVel+rne+sensors", "collision", "constraint rows", "Newton total",
         "implicitfast+advance", "obs+info epilogue", "  Newton: Hessian (MFMA)", "  Newton: Cholesky+park", "  Newton: tri. solves",
         "  Newton: line search", "  Newton: move+constraint update", "  implicitfast: factor + solve (rest = advance)"]
tot = acc[:10].sum()
print(f"mean wave lifetime {tot:.0f} cycles per control step (4 substeps), diagnostic build, {N} envs resident")
for i in list(range(10)) + [10, 11, 12, 13, 14, 15]:
    print(f"  {names[i]:34s} {acc[i]:10.0f} cycles  {100*acc[i]/tot:5.1f} %")
