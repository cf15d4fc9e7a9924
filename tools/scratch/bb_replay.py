import numpy as np, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../.."))
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.model import get_field
from oracle.oracle import Oracle
cfg = make_config("humanoid_p_v0", random=PARITY_RANDOM)
cm = compile_model(cfg); b = cm.blob
gt = np.array(b.geom_type[:b.ngeom])
bb = {(int(b.pair_geom1[p]), int(b.pair_geom2[p])) for p in range(b.npair)}
bb = {k for k in bb if gt[k[0]] == 6 and gt[k[1]] == 6}
o = Oracle(cm)
q0 = np.array(get_field(b, "init_qpos")[:b.nq])
rng = np.random.default_rng(1)
states = []
for _ in range(6000):
    q = q0.copy(); q[7:] += rng.uniform(-1.5, 1.5, b.nq - 7)
    o.reset(q); o.forward(); c = o.contacts()
    if len(c) and any((int(r[9]), int(r[7])) in bb for r in c): states.append(q)
n = len(states)
env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
env.reset()
for mode in (1, 0):
    env.engine.set_param("boxbox_mode", np.array([float(mode)], dtype=np.float32))
    o.set_boxbox_mpr(not mode)
    qv1, nc, ne = [], [], []
    for q in states:
        o.reset(q); o.control_step(np.zeros(b.nu)); qv1.append(o.qvel.copy()); nc.append(o.ncon); ne.append(o.nefc)
    qv1 = np.array(qv1)
    env.set_state(np.array(states), np.zeros((n, b.nv)), np.zeros((n, b.nv)))
    env.step(torch.zeros((n, b.nu), dtype=torch.float32, device=env.device))
    qv = env.get_data().qvel.cpu().numpy().astype(np.float64)
    ev = np.abs(qv - qv1).max(axis=1); mag = np.abs(qv1).max(axis=1)
    print("mode", mode, "n", n, "median ev", np.median(ev), "q90", np.quantile(ev, .9), "median |qv|", np.median(mag), "rel median", np.median(ev / (1 + mag)), "rel q90", np.quantile(ev/(1+mag), .9))
    print("  ncon max", max(nc), "nefc max", max(ne), env.solver_stats())
    idx = np.argsort(-ev)[:5]
    print("  worst", [(int(i), float(ev[i]), float(mag[i]), nc[i], ne[i]) for i in idx])
