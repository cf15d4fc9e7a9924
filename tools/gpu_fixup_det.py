"""Diagnostic: is the fix-up path independent of how flagged envs group into the 64-flag chunks?  Fleet vs shard, p_v3 dense experiment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cosim_amd.batched_env import BatchedEnv
from cosim_amd.compile import compile_model
from cosim_amd.config import make_config
from bench import synthetic_actions
env_id, n = "flamingo_p_v3", 4096
cfg = make_config(env_id, terrain="flat", num_envs=n, seed=1234)
cm = compile_model(cfg)
fleet = BatchedEnv(cfg, num_envs=n, seed=1234, auto_reset=True, gain_noise=0.1, compiled=cm)
lo = (n * 5) // 7
shard = BatchedEnv(cfg, num_envs=64, seed=1234, auto_reset=True, gain_noise=0.1, env_id0=lo, compiled=cm)
acts = synthetic_actions(n, 0, 60, fleet.action_dim, fleet.device)
cmd = np.array([0.5, 0.0, 0.0, 0.0], dtype=np.float32)
fleet.receive_user_command(cmd); shard.receive_user_command(cmd)
sf, _ = fleet.reset(); ss, _ = shard.reset()
def meta(e):
    buf = torch.zeros((e.num_envs, 16), dtype=torch.float32, device=e.device)
    e.engine.get("meta", buf.data_ptr(), e._stream()); torch.cuda.synchronize()
    return buf.view(torch.int32).cpu().numpy()
for t in range(60):
    sf, _, _, _ = fleet.step(acts[t]); ss, _, _, _ = shard.step(acts[t, lo:lo + 64].contiguous())
    qf, qs = fleet.get_data().qpos[lo:lo + 64].clone(), shard.get_data().qpos.clone()
    if not torch.equal(qf, qs):
        bad = ((qf - qs).abs().max(dim=1).values > 0).nonzero().flatten().cpu().numpy()
        mf, ms = meta(fleet)[lo:lo + 64], meta(shard)
        print("step", t, "envs differing (local ids)", bad, "max diff", float((qf - qs).abs().max()))
        print("  fix-up counts fleet", mf[bad, 12], "shard", ms[bad, 12], "| any fix-ups in the shard so far", ms[:, 12].sum(), "fleet slice", mf[:, 12].sum())
        print("  flagged neighbours in the fleet's chunk(s):", [int(x) for x in np.nonzero(meta(fleet)[(lo // 64) * 64:(lo // 64) * 64 + 128, 12])[0]])
        break
else:
    print("no difference in 60 steps; fix-ups fleet slice", meta(fleet)[lo:lo + 64, 12].sum())
