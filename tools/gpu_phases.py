#!/usr/bin/env python3
"""Per-phase shader-clock profile of the step kernel (diagnostic build with s_memtime stamps; shares, not run times).

    python tools/gpu_phases.py [workload] [settle_steps]      workloads: bench.py's names that have a diagnostic kernel
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.config import make_config
from bench import WORKLOADS, synthetic_actions
wl = sys.argv[1] if len(sys.argv) > 1 else "light_flat"
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 300
robot, terrain, hmap, N = WORKLOADS[wl]
poscmd = wl == "humanoid_stairs"
cfg = make_config(robot, terrain=terrain, num_envs=N, seed=1234, height_map=hmap, position_command=poscmd)
if poscmd:
    cfg["observation"]["command_dim"] = 2
env = BatchedEnv(cfg, num_envs=N, seed=1234, auto_reset=True, gain_noise=0.1)
K = 20
acts = synthetic_actions(N, 0, settle + K + 20, env.action_dim, env.device)
env.receive_user_command(np.array([0.5, 0.0, 0.0, 0.0], dtype=np.float32)[:max(env.command_dim, 1)])
env.reset()
for t in range(settle):
    env.step(acts[t])
torch.cuda.synchronize()
acc = np.zeros(32)
for t in range(settle, settle + K):
    a = acts[t].contiguous()
    acc += env.engine.profile_step(a.data_ptr(), env._cmd_ptr(), env.state.data_ptr(), env.terminated.data_ptr(), env.truncated.data_ptr())
acc /= K
names = ["prologue", "kinematics", "comPos+cdof", "crb (M)", "comVel+rne+sensors", "collision", "constraint rows", "Newton total",
         "implicitfast+advance", "obs+info epilogue", "  Newton: Hessian (MFMA + contact tree pass)", "  Newton: Cholesky+park", "  Newton: tri. solves",
         "  Newton: line search", "  Newton: move+constraint update", "  implicitfast: factor + solve (rest = advance)"]
tot = acc[:10].sum()
st = env.solver_stats()
print(f"{wl}: mean wave lifetime {tot:.0f} cycles per control step ({int(env.cm.blob.frame_skip)} substeps), diagnostic build, {N} envs; "
      f"contact slots {env.engine.query('contact_slots')}, max contacts seen {st['max_contacts']}, dropped {st['dropped_contacts']}")
for i in list(range(10)) + [10, 11, 12, 13, 14, 15]:
    print(f"  {names[i]:46s} {acc[i]:12.0f} cycles  {100*acc[i]/tot:5.1f} %")
if acc[19] > 0:
    print(f"  (collision - staged prism walk = cooperative hull walk + robot-robot pairs: {acc[5]-acc[16]-acc[17]-acc[18]:.0f} cycles per step)")
    sub = int(env.cm.blob.frame_skip)
    print(f"  heightfield narrowphase per substep: slowest-lane MPR iterations summed over full batches {acc[19]/sub:.0f}, {acc[20]/sub:.1f} probe batches, "
          f"{acc[22]/sub:.1f} full-MPR batches (inside mpr_penetration {acc[21]/sub:.0f} cycles, set-up before it {acc[23]/sub:.0f}); cycles: sub-grids {acc[16]/sub:.0f}, probe passes {acc[17]/sub:.0f}, "
          f"full batches {acc[18]/sub:.0f}")
elif acc[16] > 0:
    print(f"  robot-robot pairs (broadphase + MPR): {acc[16]:.0f} cycles = {100*acc[16]/tot:.1f} % of the step")
    print(f"  contact-twist Hessian: per-body matrices {acc[17]:.0f} cycles ({100*acc[17]/tot:.1f} %), tree pass {acc[18]:.0f} cycles ({100*acc[18]/tot:.1f} %)")
sub = int(env.cm.blob.frame_skip)
if acc[19] > 0 and acc[27] > 0:
    print(f"  cooperative hull walk per substep: {acc[27]/sub:.1f} geoms, exact boxes {acc[24]/sub:.0f} cycles, {acc[25]/sub:.1f} MPR runs ({acc[28]/sub:.2f} hits, "
          f"{acc[29]/max(acc[25],1):.1f} refinement iterations per run), {acc[26]/max(acc[25],1):.0f} cycles per run; MPR {acc[26]:.0f} + boxes {acc[24]:.0f} cycles per step = "
          f"{100*(acc[26]+acc[24])/tot:.1f} %")
elif acc[24] > 0:
    print(f"  robot-robot pairs with a hull, per substep: {acc[28]/sub:.1f} past the bounding spheres, {acc[24]/sub:.2f} MPR runs ({acc[25]/sub:.2f} hits), "
          f"{acc[26]/max(acc[24],1):.1f} refinement iterations per run, {acc[27]/max(acc[24],1):.0f} cycles per run ({acc[27]:.0f} per step = {100*acc[27]/tot:.1f} %)")
