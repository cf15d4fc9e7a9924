#!/usr/bin/env python3
"""GPU-side debugging probe: per-stage comparison of the HIP engine with the CPU oracle.

Writes a report under gpurun_out/.  Not part of the product or of the test suite; it is the
tool used while bringing the kernel up (one `gpurun` call -> which stage diverges).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from cosim_amd.batched_env import BatchedEnv  # noqa: E402
from cosim_amd.compile import compile_model  # noqa: E402
from cosim_amd.config import PARITY_RANDOM, make_config  # noqa: E402
from cosim_amd.model import get_field  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def tri_to_dense(tri, nv):
    M = np.zeros((nv, nv))
    e = 0
    for r in range(nv):
        for c in range(r + 1):
            M[r, c] = M[c, r] = tri[e]
            e += 1
    return M


def main():
    import torch
    out = open(os.path.join("gpurun_out", "probe.txt"), "w")

    def P(*a):
        print(*a, file=out, flush=True)
        print(*a, flush=True)

    cfg = make_config("flamingo_light_v1", random=PARITY_RANDOM, num_envs=4)
    cm = compile_model(cfg)
    env = BatchedEnv(cfg, num_envs=4, auto_reset=False, compiled=cm)
    P("lds_bytes", env.engine.query("lds_bytes"), "state_dim", env.state_dim, "stride", env.engine.query("state_stride"))
    state, _ = env.reset()
    torch.cuda.synchronize()
    P("reset state[0]:", state[0].cpu().numpy().round(4))
    nv, nb = 18, 14
    o = Oracle(cm)
    q0 = np.array(get_field(cm.blob, "init_qpos")[:19])
    o.reset(q0)
    o.forward()
    D = env.engine.debug_forward(0)
    P("ncon nefc ne nf nl cost grad ngen niter cost grad:", D[:11])
    P("oracle ncon nefc ne nf nl niter:", o.ncon, o.nefc, o.ne, o.nf, o.nl, o.solver_niter)
    xpos = D[64:64 + nb * 3].reshape(nb, 3)
    P("xpos maxdiff", np.abs(xpos - o.xpos).max())
    xq = D[192:192 + nb * 4].reshape(nb, 4)
    P("xquat maxdiff", np.abs(xq - o.xquat).max())
    M = tri_to_dense(D[512:512 + nv * (nv + 1) // 2], nv)
    P("M maxdiff", np.abs(M - o.M).max(), "rel", np.abs(M - o.M).max() / np.abs(o.M).max())
    P("qfrc_smooth diff", np.abs(D[1100:1100 + nv] - o.qfrc_smooth).max(), "bias diff", np.abs(D[1140:1140 + nv] - o.qfrc_bias).max())
    P("cdof diff", np.abs(D[1200:1200 + nv * 6].reshape(nv, 6) - o.cdof).max())
    P("rtype", D[1400:1464].astype(int))
    P("contacts dist gpu", D[1720:1734], "\n oracle", o.contacts()[:, 0])
    P("qacc gpu", D[1000:1000 + nv].round(3))
    P("qacc orc", o.qacc.round(3))
    P("qacc maxdiff", np.abs(D[1000:1000 + nv] - o.qacc).max())
    P("qfrc_constraint diff", np.abs(D[1040:1040 + nv] - o.qfrc_constraint).max())
    P("sensor quat", D[16:20], o.sensor_quat, "gyro", D[20:23], "vel", D[24:27])

    # trajectory parity, zero action
    N = 4
    act = torch.zeros((N, 4), device=env.device)
    T = 300
    err = []
    t0 = time.time()
    for t in range(T):
        env.step(act)
        o.control_step(np.zeros(4))
        if t % 10 == 9 or t < 5:
            d = env.get_data()
            torch.cuda.synchronize()
            qg = d.qpos[0].cpu().numpy().astype(np.float64)
            err.append((t, np.abs(qg - o.qpos).max(), np.abs(d.qvel[0].cpu().numpy() - o.qvel).max()))
    P("zero-action trajectory (step, |dqpos|max, |dqvel|max):")
    for e in err:
        P("  ", e)
    P("gpu qpos", qg.round(4))
    P("orc qpos", o.qpos.round(4))
    P("state[0]", env.state[0].cpu().numpy().round(3))
    P("nan count", int(torch.isnan(env.state).sum().item()))

    # sinusoid actions, fresh envs
    env.reset()
    o.reset(q0)
    errs = []
    for t in range(500):
        a = 0.25 * np.sin(2 * np.pi * 0.5 * 0.02 * t + np.array([0.0, 1.0, 2.0, 3.0]))
        act = torch.tensor(np.tile(a, (N, 1)), dtype=torch.float32, device=env.device)
        env.step(act)
        o.control_step(a)
        if t % 25 == 24:
            d = env.get_data()
            torch.cuda.synchronize()
            qg = d.qpos[0].cpu().numpy().astype(np.float64)
            errs.append((t, np.abs(qg[7:] - o.qpos[7:]).max(), np.sqrt(np.mean((qg[7:] - o.qpos[7:]) ** 2))))
    P("sinusoid trajectory (step, max joint err, rms joint err):")
    for e in errs:
        P("  ", e)

    # throughput
    for N in (1024, 4096, 16384):
        cfgb = make_config("flamingo_light_v1", num_envs=N)
        envb = BatchedEnv(cfgb, num_envs=N, auto_reset=True, compiled=None)
        envb.reset()
        act = torch.zeros((N, 4), device=envb.device)
        for _ in range(20):
            envb.step(act)
        torch.cuda.synchronize()
        t0 = time.time()
        K = 100
        for _ in range(K):
            envb.step(act)
        torch.cuda.synchronize()
        dt = time.time() - t0
        P(f"N={N}: {K} steps in {dt:.3f}s -> {N * K / dt:.0f} env-steps/s, {dt / K * 1e3:.3f} ms/step; nan={int(torch.isnan(envb.state).sum().item())}")
        envb.close()


if __name__ == "__main__":
    main()
