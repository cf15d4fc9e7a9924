#!/usr/bin/env python3
"""GPU probe: heightfield (prism MPR) contacts of the HIP engine vs the oracle on states scattered over a terrain."""
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


def run(env_id, terrain, spots=12, steps=40):
    cfg = make_config(env_id, terrain=terrain, random=PARITY_RANDOM)
    cm = compile_model(cfg)
    b = cm.blob
    rng = np.random.default_rng(11)
    o = Oracle(cm)
    q0 = np.array(get_field(b, "init_qpos")[:b.nq])
    R = dict(qpos=[], qvel=[], warm=[], act=[], qpos1=[], qvel1=[], ncon=[])
    for spot in range(spots):
        q = q0.copy()
        q[0:2] = rng.uniform(-100, 100, size=2)
        yaw = rng.uniform(-np.pi, np.pi)
        q[3:7] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
        q[2] = q0[2] + (10.0 - o.ray_down(q[0], q[1], 10.0)) + 0.02
        o.reset(q)
        for t in range(steps):
            a = np.clip(0.1 * rng.normal(size=b.nu), -1, 1)
            R["qpos"].append(o.qpos.copy()); R["qvel"].append(o.qvel.copy()); R["warm"].append(o.qacc_warmstart.copy()); R["act"].append(a)
            o.control_step(a)
            R["qpos1"].append(o.qpos.copy()); R["qvel1"].append(o.qvel.copy()); R["ncon"].append(o.ncon)
    R = {k: np.array(v) for k, v in R.items()}
    n = len(R["qpos"])
    env = BatchedEnv(cfg, num_envs=n, auto_reset=False, compiled=cm)
    env.reset()
    env.set_state(R["qpos"], R["qvel"], R["warm"])
    nm = nbad = ncnt = 0
    for w in range(n):
        o.reset(R["qpos"][w], R["qvel"][w])
        o.forward()
        oc = o.contacts()
        dbg = env.engine.debug_forward(int(w))
        nc = min(int(dbg[0]), 16)
        base = R["qpos"][w][:3].copy(); base[2] = 0.0
        gl = sorted([(int(dbg[1900 + i]) & 255, float(dbg[1720 + i]), dbg[1740 + 3 * i:1743 + 3 * i] + base, dbg[1920 + 3 * i:1923 + 3 * i].copy()) for i in range(nc)], key=lambda c: (c[0], round(c[2][0], 3), round(c[2][1], 3)))
        ol = sorted([(int(c[7]), c[0], c[1:4], c[4:7]) for c in oc], key=lambda c: (c[0], round(c[2][0], 3), round(c[2][1], 3)))
        if len(gl) != len(ol) or any(a[0] != b_[0] for a, b_ in zip(gl, ol)):
            ncnt += 1
            if ncnt <= 6:
                print(f"  state {w}: contact sets differ: oracle {[(c[0], round(c[1], 5)) for c in ol]} gpu {[(c[0], round(c[1], 5)) for c in gl]}")
            continue
        for a, c in zip(gl, ol):
            nm += 1
            ed, en, ep = abs(a[1] - c[1]), np.abs(a[3] - c[3]).max(), np.abs(a[2] - c[2]).max()
            if ed > 2e-5 or en > 2e-3 or ep > 2e-3:
                nbad += 1
                if nbad <= 12:
                    print(f"  state {w} geom {a[0]}: dist {c[1]:.5f}/{a[1]:.5f} n {np.round(c[3], 3)}/{np.round(a[3], 3)} pos {np.round(c[2], 3)}/{np.round(a[2], 3)}")
    env.step(torch.tensor(R["act"], dtype=torch.float32, device=env.device))
    d = env.get_data()
    qv = d.qvel.cpu().numpy().astype(np.float64)
    ev = np.abs(qv - R["qvel1"]).max(axis=1)
    print(f"{env_id} {terrain}: states {n}, contact-set mismatches {ncnt}, contacts compared {nm}, off {nbad}; replay |dqvel| median {np.median(ev):.2e} p90 {np.quantile(ev, .9):.2e} p95 {np.quantile(ev, .95):.2e} max {ev.max():.2e}")
    env.close()


if __name__ == "__main__":
    run("flamingo_light_v1", "rocky_hard")
    run("w4_p_v2", "rocky_hard")
    run("humanoid_p_v0", "rocky_hard")
