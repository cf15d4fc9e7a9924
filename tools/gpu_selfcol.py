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
    # contact-level comparison (MPR parity) on every state that starts with a self contact in the oracle
    env.set_state(R["qpos"], R["qvel"], R["warm"])
    worst = dict(dist=0.0, nrm=0.0, pos=0.0)
    nmatch = nmiss = 0
    for w in range(n):
        o.reset(R["qpos"][w], R["qvel"][w])
        o.forward()
        oc = o.contacts()
        if not len(oc) or not (oc[:, 9] >= 0).any():
            continue
        dbg = env.engine.debug_forward(int(w))
        nc = min(int(dbg[0]), 16)
        gpu = {}
        for i in range(nc):
            gg = int(dbg[1900 + i])
            if (gg >> 8) - 1 >= 0:
                gpu[((gg >> 8) - 1, gg & 255)] = (dbg[1720 + i], dbg[1740 + 3 * i:1743 + 3 * i].copy(), dbg[1920 + 3 * i:1923 + 3 * i].copy())
        base = R["qpos"][w][:3].copy(); base[2] = 0.0           # the engine works in a base-relative frame
        for c in oc[oc[:, 9] >= 0]:
            key = (int(c[9]), int(c[7]))
            if key not in gpu:
                nmiss += 1
                print(f"   state {w}: pair {key} dist {c[0]:.2e} missing on the GPU (gpu self pairs: {list(gpu)})")
                continue
            nmatch += 1
            gd, gp, gn = gpu[key]
            ed, en, ep = abs(gd - c[0]), np.abs(gn - c[4:7]).max(), np.abs(gp + base - c[1:4]).max()
            if max(ed, en * 1e-2, ep * 1e-1) > 1e-4:
                print(f"   state {w} pair {key}: dist {c[0]:.5f}/{gd:.5f} n {np.round(c[4:7],3)}/{np.round(gn,3)} pos {np.round(c[1:4]-base,4)}/{np.round(gp,4)}")
            worst["dist"] = max(worst["dist"], ed); worst["nrm"] = max(worst["nrm"], en); worst["pos"] = max(worst["pos"], ep)
    print(f"   MPR contacts matched {nmatch}, missing {nmiss}; max |d dist| {worst['dist']:.2e} |d normal| {worst['nrm']:.2e} |d pos| {worst['pos']:.2e}")
    env.close()


if __name__ == "__main__":
    run("humanoid_p_v0", 400, 0.6)
    run("flamingo_p_v3", 300, 0.9)
    run("w4_p_v2", 300, 0.9)
