"""Diagnostic: one-control-step replay of drop poses with 15-34 plane contacts, with and without the fix-up kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.model import get_field
from oracle.oracle import Oracle
cfg = make_config("flamingo_light_v1", random=PARITY_RANDOM, num_envs=4)
cm = compile_model(cfg)
q0 = np.array(get_field(cm.blob, "init_qpos")[:cm.blob.nq])
o = Oracle(cm)
rng = np.random.default_rng(3)
R = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[], ncon=[])
for trial in range(200):
    q = q0.copy()
    quat = rng.normal(size=4)
    q[2] = rng.uniform(0.05, 0.25)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.uniform(-0.3, 0.3, size=q.size - 7)
    o.reset(q)
    for t in range(2):
        a = 0.3 * np.sin(0.3 * t + np.arange(4))
        pre = (o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy())
        o.control_step(a)
        R["qpos"].append(pre[0]); R["qvel"].append(pre[1]); R["warm"].append(pre[2]); R["act"].append(a)
        R["qpos1"].append(o.qpos.copy()); R["qvel1"].append(o.qvel.copy()); R["ncon"].append(o.ncon)
R = {k: np.array(v) for k, v in R.items()}
big = R["ncon"] > 14
print("states", len(big), "big", big.sum(), "max", R["ncon"].max())
n = len(big)
for fixup in (True, False):
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
    if not fixup:
        env.engine.set_param("fixup", np.array([0.0]))
    env.reset()
    env.set_state(R["qpos"], R["qvel"], R["warm"])
    env.step(torch.tensor(R["act"], dtype=torch.float32, device=env.device))
    d = env.get_data()
    qp, qv = d.qpos.cpu().numpy().astype(np.float64), d.qvel.cpu().numpy().astype(np.float64)
    st = env.solver_stats()
    ev = np.abs(qv - R["qvel1"]).max(axis=1)
    dv = np.abs(R["qvel1"] - R["qvel"]).max(axis=1)
    ep = np.abs(qp - R["qpos1"]).max(axis=1)
    print("fixup", fixup, {k: st[k] for k in ("dropped_contacts", "fixup_steps", "max_contacts", "nan_resets")})
    for name, m in (("big", big), ("small", ~big)):
        print("  ", name, "ev med %.2e q95 %.2e max %.2e | rel med %.2e q95 %.2e | ep max %.2e" % (
            np.median(ev[m]), np.quantile(ev[m], .95), ev[m].max(), np.median(ev[m] / (dv[m] + 1)), np.quantile(ev[m] / (dv[m] + 1), .95), ep[m].max()))
    env.close()
