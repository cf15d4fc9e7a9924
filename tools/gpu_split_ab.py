"""Diagnostic: the split pipeline (narrowphase kernel + per-substep solver kernel) against the fused kernel on identical states."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import synthetic_actions
N = 256
cfg = make_config("humanoid_p_v0", terrain="stairs_up_hard", num_envs=N, seed=1234, position_command=True)
cfg["observation"]["command_dim"] = 2
a = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
b = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1, compiled=a.cm)
b.engine.set_param("split", np.array([0.0]))
print("split:", a.engine.query("split"), b.engine.query("split"))
acts = synthetic_actions(N, 0, 300, a.action_dim, a.device)
tgt = np.random.default_rng(0).uniform(-3, 3, size=(N, 2)).astype(np.float32)
a.receive_user_command(tgt); b.receive_user_command(tgt)
sa, _ = a.reset(); sb, _ = b.reset()
print("reset equal", torch.equal(sa, sb))
first = None
for t in range(300):
    sa, ta, ca, ia = a.step(acts[t]); sb, tb, cb, ib = b.step(acts[t])
    if first is None and not torch.equal(sa, sb):
        first = t
        d = (sa - sb).abs().max(dim=1).values
        print("first difference at step", t, "envs differing", int((d > 0).sum()), "max", float(d.max()))
        qa, qb = a.get_data().qpos.clone(), b.get_data().qpos.clone()
        print("  qpos max diff", float((qa - qb).abs().max()))
    if t in (0, 1, 5, 20, 100, 299):
        print(t, "state equal", torch.equal(sa, sb), "stats split", {k: a.solver_stats()[k] for k in ("dropped_contacts", "max_contacts", "nan_resets")},
              "fused", {k: b.solver_stats()[k] for k in ("dropped_contacts", "max_contacts", "nan_resets")})
