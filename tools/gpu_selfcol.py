#!/usr/bin/env python3
"""GPU probe: one-control-step replay along oracle trajectories that include robot-robot (self) contacts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import PARITY_RANDOM, make_config
from cosim_amd.model import get_field
from oracle.oracle import Oracle


def run(env_id, steps, amp, seed=0):
    cfg = make_config(env_id, random=PARITY_RANDOM)
    cm = compile_model(cfg)
    b = cm.blob
    o = Oracle(cm)
    o.reset(np.array(get_field(b, "init_qpos")[:b.nq]))
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0, 6.28, b.nu)
    R = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[], nefc=[], nself=[], ncon=[])
    for t in range(steps):
        a = np.clip(amp * np.sin(2 * np.pi * 0.5 * t * 0.02 + phi), -1, 1)
        R["qpos"].append(o.qpos.copy()); R["qvel"].append(o.qvel.copy()); R["warm"].append(o.qacc_warmstart.copy()); R["act"].append(a)
        o.control_step(a)
        c = o.contacts()
        R["nself"].append(int((c[:, 9] >= 0).sum()) if len(c) else 0)
        R["ncon"].append(o.ncon)
        R["qpos1"].append(o.qpos.copy()); R["qvel1"].append(o.qvel.copy()); R["nefc"].append(o.nefc)
        if o.bad:
            print("oracle went bad at", t)
            break
    R = {k: np.array(v) for k, v in R.items()}
    n = len(R["qpos"])
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
    env.reset()
    env.set_state(R["qpos"], R["qvel"], R["warm"])
    env.step(torch.tensor(R["act"], dtype=torch.float32, device=env.device))
    d = env.get_data()
    qp, qv = d.qpos.cpu().numpy().astype(np.float64), d.qvel.cpu().numpy().astype(np.float64)
    ev = np.abs(qv - R["qvel1"]).max(axis=1)
    ep = np.abs(qp - R["qpos1"]).max(axis=1)
    ok = R["nefc"] <= 110
    sc = R["nself"] > 0
    print(f"{env_id}: steps {n}, with self contacts {sc.sum()}, nefc max {R['nefc'].max()}, ncon max {R['ncon'].max()}")
    for name, m in (("no self contact", ok & ~sc), ("self contact", ok & sc)):
        if m.sum():
            print(f"   {name:16s} n={m.sum():4d}  |dqvel| median {np.median(ev[m]):.2e} p90 {np.quantile(ev[m], 0.9):.2e} max {ev[m].max():.2e}   |dqpos| max {ep[m].max():.2e}")
    # contact-level comparison on the worst self-contact state
    if sc.sum():
        idx = np.nonzero(ok & sc)[0]
        w = idx[np.argmax(ev[idx])]
        o.reset(R["qpos"][w], R["qvel"][w])
        o.forward()
        oc = o.contacts()
        env.set_state(R["qpos"], R["qvel"], R["warm"])
        dbg = env.engine.debug_forward(int(w))
        nc = int(dbg[0])
        print(f"   worst self-contact state {w}: oracle ncon {len(oc)} gpu ncon {nc}")
        for i in range(len(oc)):
            print("     oracle", f"g1 {int(oc[i,9]):3d} g2 {int(oc[i,7]):3d} dist {oc[i,0]: .5f} pos", np.round(oc[i, 1:4], 4), "n", np.round(oc[i, 4:7], 3))
        for i in range(min(nc, 16)):
            gg = int(dbg[1800 + i])
            print("     gpu   ", f"g1 {(gg >> 8) - 1:3d} g2 {gg & 255:3d} dist {dbg[1720+i]: .5f} pos", np.round(dbg[1740 + 3 * i:1743 + 3 * i], 4), "n", np.round(dbg[1820 + 3 * i:1823 + 3 * i], 3))
    env.close()


if __name__ == "__main__":
    run("humanoid_p_v0", 400, 0.6)
    run("flamingo_p_v3", 300, 0.9)
    run("w4_p_v2", 300, 0.9)
