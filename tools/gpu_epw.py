#!/usr/bin/env python3
"""GPU probe: the two-environments-per-wave kernel against the one-per-wave kernel, step by step."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import PARITY_RANDOM, make_config

def lanes(g, tag):
    for it in range(3):
        for nm, base in (("alpha", 5000), ("cost", 5300), ("gauss", 5600)):
            v = g[base + it * 64: base + it * 64 + 64]
            if np.unique(v).size > 1:
                print(f"     {tag} iter {it} {nm} NOT uniform:", np.unique(v)[:6], "lanes differing from lane 0:", np.nonzero(v != v[0])[0][:20])


cfg = make_config("flamingo_light_v1", random=PARITY_RANDOM)
N = 8
os.environ.pop("COSIM_ENVS_PER_WAVE", None)
e1 = BatchedEnv(cfg, num_envs=N, auto_reset=False)
e2 = BatchedEnv(cfg, num_envs=N, auto_reset=False)
e2.engine.set_param("envs_per_wave", np.array([2.0]))
e1.reset(); e2.reset()
g1, g2 = e1.engine.debug_forward(0), e2.engine.debug_forward(0)
names = {0: "ncon nefc ne nf nl cost gradnorm ngen niter cost2 gradnorm2", 16: "sens", 64: "xpos", 192: "xquat", 512: "M tri", 1000: "qacc", 1040: "qcon",
         1080: "qacc implicit", 1100: "qsm", 1140: "bias", 1200: "cdof", 1400: "rtype", 1464: "rD", 1528: "raref", 1592: "rpos", 1656: "Jaref", 1720: "cdist",
         1740: "cpos", 1800: "rowf", 2048: "J"}
keys = sorted(names)
for i, k in enumerate(keys):
    hi = keys[i + 1] if i + 1 < len(keys) else 2048 + 54 * 18
    d = np.abs(g1[k:hi] - g2[k:hi])
    print(f"  dbg[{k}:{hi}] {names[k]:40s} max diff {d.max():.3e}" + (f"   epw1 {g1[k:k+11]}  epw2 {g2[k:k+11]}" if k == 0 else ""))
def warm(e):
    w = torch.empty((N, 18), device=e.device)
    e.engine.get("qacc_warmstart", w.data_ptr(), None)
    torch.cuda.synchronize()
    return w
for e in (e1, e2):
    e.engine.set_param("debug_substeps", np.array([1.0]))
for t in range(6):
    a = torch.zeros((N, 4), device=e1.device)
    e1.step(a); e2.step(a)
    d1, d2 = e1.get_data(), e2.get_data()
    print(f"substep {t}: |dqpos| {(d1.qpos - d2.qpos).abs().max().item():.3e} |dqvel| {(d1.qvel - d2.qvel).abs().max().item():.3e} |dwarm| {(warm(e1) - warm(e2)).abs().max().item():.3e}",
          "stats", e1.solver_stats()["newton_iters"], e2.solver_stats()["newton_iters"], "rows", e1.solver_stats()["rows"], e2.solver_stats()["rows"])
    # same state into both, compare the next forward pass
    e2.set_state(d1.qpos.cpu().numpy(), d1.qvel.cpu().numpy(), warm(e1).cpu().numpy())
    g1, g2 = e1.engine.debug_forward(0), e2.engine.debug_forward(0)
    lanes(g1, "epw1"); lanes(g2, "epw2")
    if abs(g1[9] - g2[9]) > 1e-3:
        from oracle.oracle import Oracle
        o = Oracle(e1.cm)
        o.reset(d1.qpos[0].cpu().numpy().astype(np.float64), d1.qvel[0].cpu().numpy().astype(np.float64))
        o.qacc_warmstart[:] = warm(e1)[0].cpu().numpy()
        o.forward()
        for it in range(4):
            b = 3100 + it * 400
            print(f"     iter {it}: cost {g1[b+390]:.5f}/{g2[b+390]:.5f} act {g1[b+391]}/{g2[b+391]}  |dL| {np.abs(g1[b:b+324]-g2[b:b+324]).max():.3e} |dsr| {np.abs(g1[b+330:b+348]-g2[b+330:b+348]).max():.3e} |dgrad| {np.abs(g1[b+360:b+378]-g2[b+360:b+378]).max():.3e}")
        print("     oracle: nefc", o.nefc, "niter", o.solver_niter, "qacc head", np.round(o.qacc[:12], 4))
    for i, k in enumerate(keys):
        hi = keys[i + 1] if i + 1 < len(keys) else 2048 + 54 * 18
        if names[k] in ("rtype", "rD", "raref", "rpos", "Jaref"):
            hi = k + 32
        d = np.abs(g1[k:hi] - g2[k:hi])
        if d.max() > 1e-4 * max(1.0, np.abs(g1[k:hi]).max()):
            print(f"     dbg {names[k]}: max diff {d.max():.3e} (scale {np.abs(g1[k:hi]).max():.3e})  head1 {np.round(g1[k:k+12], 4)} head2 {np.round(g2[k:k+12], 4)}")
