#!/usr/bin/env python3
"""Replay probe: states recorded along an oracle trajectory are loaded into a GPU batch (one env per recorded
state) and advanced by ONE control step; compares against the oracle's own next state.  Localises per-stage
differences for the worst env with the debug kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from cosim_amd.batched_env import BatchedEnv  # noqa: E402
from cosim_amd.compile import compile_model  # noqa: E402
from cosim_amd.config import PARITY_RANDOM, make_config  # noqa: E402
from cosim_amd.model import get_field  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    import torch
    out = open(os.path.join("gpurun_out", "probe2.txt"), "w")

    def P(*a):
        print(*a, file=out, flush=True)
        print(*a, flush=True)

    T = 400
    cfg = make_config("flamingo_light_v1", random=PARITY_RANDOM, num_envs=T)
    cm = compile_model(cfg)
    o = Oracle(cm)
    q0 = np.array(get_field(cm.blob, "init_qpos")[:19])
    o.reset(q0)
    rec = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[], ncon=[])
    for t in range(T):
        a = 0.25 * np.sin(2 * np.pi * 0.5 * 0.02 * t + np.array([0.0, 1.0, 2.0, 3.0]))
        rec["qpos"].append(o.qpos.copy()); rec["qvel"].append(o.qvel.copy()); rec["warm"].append(o.qacc_warmstart.copy())
        rec["act"].append(a)
        o.control_step(a)
        rec["qpos1"].append(o.qpos.copy()); rec["qvel1"].append(o.qvel.copy()); rec["ncon"].append(o.ncon)
    R = {k: np.array(v) for k, v in rec.items()}
    env = BatchedEnv(cfg, num_envs=T, auto_reset=False, compiled=cm)
    env.reset()
    env.set_state(R["qpos"], R["qvel"], R["warm"])
    act = torch.tensor(R["act"], dtype=torch.float32, device=env.device)
    env.step(act)
    d = env.get_data()
    torch.cuda.synchronize()
    qp = d.qpos.cpu().numpy().astype(np.float64)
    qv = d.qvel.cpu().numpy().astype(np.float64)
    ep = np.abs(qp - R["qpos1"]).max(axis=1)
    ev = np.abs(qv - R["qvel1"]).max(axis=1)
    P("one-control-step replay over", T, "states: max |dqpos|", ep.max(), "max |dqvel|", ev.max())
    order = np.argsort(-ev)[:12]
    P("worst (t, dqpos, dqvel, ncon_after):")
    for i in order:
        P("  ", i, ep[i], ev[i], R["ncon"][i])
    P("median dqvel", np.median(ev), "90%", np.quantile(ev, 0.9), "99%", np.quantile(ev, 0.99))
    # localise the worst one with the debug forward (first substep only)
    w = int(order[0])
    for w in [int(order[0]), 170, 171]:
        o.reset(R["qpos"][w], R["qvel"][w])
        o.view("qacc_warmstart")[:] = R["warm"][w]
        # control torque
        m = cm.blob
        a = R["act"][w]
        tq = np.zeros(4)
        for u in range(4):
            s, g = m.ctl_scale[u], m.ctl_gear[u]
            q, qd = o.qpos[m.ctl_qadr[u]] * g, o.qvel[m.ctl_dadr[u]] * g
            t_ = m.ctl_kd[u] * (a[u] * s - qd) if m.ctl_velmode[u] else m.ctl_kp[u] * (a[u] * s - q) + m.ctl_kd[u] * (0 - qd)
            tq[u] = np.clip(t_ * m.ctl_gamma[u], -m.ctl_maxtq[u], m.ctl_maxtq[u])
        o.view("ctrl")[:] = 0.0  # the debug kernel runs with zero actuation
        o.forward()
        env.set_state(R["qpos"], R["qvel"], R["warm"])
        D = env.engine.debug_forward(w)
        P(f"--- env {w}: gpu ncon nefc ne nf nl", D[:5], "niter", D[8], "| oracle", o.ncon, o.nefc, o.ne, o.nf, o.nl, o.solver_niter)
        P("   gpu contact dist", D[1720:1734][: int(D[0])], "\n   orc contact dist", o.contacts()[:, 0], "geoms", o.contacts()[:, 7])
        P("   qacc diff (ctrl=0 both):", np.abs(D[1000:1018] - o.qacc).max())


if __name__ == "__main__":
    main()
