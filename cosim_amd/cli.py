"""Headless runner: the reference's GUI + ``Tester`` session as one command.

    python -m cosim_amd.cli --env flamingo_light_v1 --num-envs 4096 --steps 1000 --command 0.5 0 0 0 \\
        --policy sinusoid | random-mlp | path/to/actor.onnx  [--terrain rocky_hard] [--push-at 200 --push 0.5 0 0] \\
        [--report report.json] [--trace-env 0]

One process per GPU: under ``torchrun`` every rank simulates its shard of ``--num-envs`` and rank 0 writes the report.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="cosim_amd.cli", description=__doc__.split("\n")[0])
    ap.add_argument("--env", default="flamingo_light_v1")
    ap.add_argument("--terrain", default="flat")
    ap.add_argument("--num-envs", type=int, default=1024, help="total over all ranks")
    ap.add_argument("--steps", type=int, default=500, help="control steps (50 Hz)")
    ap.add_argument("--policy", default="sinusoid", help="sinusoid | random-mlp | <file.onnx>")
    ap.add_argument("--lstm", action="store_true", help="the ONNX file is an LSTM policy with h_in / c_in inputs")
    ap.add_argument("--hidden-dim", type=int, default=256, help="h_in_dim = c_in_dim of an LSTM policy")
    ap.add_argument("--command", type=float, nargs="*", default=[0.5, 0.0, 0.0, 0.0])
    ap.add_argument("--position-command", action="store_true")
    ap.add_argument("--max-duration", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--push-at", type=int, default=-1, help="control step at which a push event fires")
    ap.add_argument("--push", type=float, nargs=3, default=[0.5, 0.0, 0.0])
    ap.add_argument("--report", default="", help="write the fleet report (JSON) here")
    ap.add_argument("--trace-env", type=int, default=-1, help="also keep the per-step info series of this local env")
    ap.add_argument("--graph", action="store_true", help="capture policy -> step -> report in a HIP graph and replay it (ONNX policies)")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args(argv)

    import torch
    from .batched_env import BatchedEnv
    from .config import make_config
    from .distributed import init_from_env, shard_range
    from .policy import build_policy, write_random_mlp
    from .reporter import FleetReporter
    from .runner import Runner, SinusoidPolicy

    rank, world = init_from_env(args.backend)
    lo, hi = shard_range(args.num_envs, rank, world)
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = make_config(args.env, terrain=args.terrain, max_duration=args.max_duration, position_command=args.position_command,
                      num_envs=hi - lo, seed=args.seed, device=dev)
    env = BatchedEnv(cfg, num_envs=hi - lo, device=dev, seed=args.seed, auto_reset=True, env_id0=lo)
    if args.policy == "sinusoid":
        policy = SinusoidPolicy(env.num_envs, env.action_dim, env.device, env_id0=lo, seed=args.seed)
    else:
        path = args.policy
        if args.policy == "random-mlp":
            path = os.path.join(tempfile.mkdtemp(prefix="cosim_policy_"), "actor.onnx")
            write_random_mlp(path, env.state_dim, env.action_dim, seed=args.seed)
        pc = {"policy": {"use_lstm": bool(args.lstm), "h_in_dim": args.hidden_dim, "c_in_dim": args.hidden_dim}}
        policy = build_policy(pc, path, num_envs=env.num_envs, device=env.device)
    if args.graph and not getattr(policy, "graph_safe", False):
        ap.error("--graph needs an ONNX policy (random-mlp or a file): the sinusoid drive keeps its clock on the host, a captured "
                 "graph would replay one frozen action")
    if args.graph and (args.push_at >= 0 or args.trace_env >= 0):
        ap.error("--graph replays one captured control step: --push-at and --trace-env need the eager loop")
    rep = FleetReporter(env, trace_env=args.trace_env if args.trace_env >= 0 else None)
    run = Runner(env, policy, reporter=rep)
    for i, v in enumerate(args.command[:env.command_dim]):
        run.update_command(i, v)

    def on_step(k, state, terminated, truncated, info):
        rep.note_done(terminated, truncated)
        if k + 1 == args.push_at:
            run.activate_push_event(np.asarray(args.push, dtype=np.float32))
        elif k == args.push_at:
            run.deactivate_push_event()
    torch.cuda.synchronize(env.device)
    t0 = time.perf_counter()
    n = run.test_graphed(args.steps) if args.graph else run.test(max_steps=args.steps, on_step=on_step)
    torch.cuda.synchronize(env.device)
    dt = time.perf_counter() - t0
    out = rep.save(args.report) if (args.report and rank == 0) else rep.summary()
    if rank == 0:
        print(json.dumps({"env": args.env, "terrain": args.terrain, "envs_total": args.num_envs, "ranks": world, "control_steps": n,
                          "env_steps_per_s_this_rank": env.num_envs * n / dt, "episodes_ended": out["episodes_ended"],
                          "metrics": {k: round(v["mean"], 5) for k, v in out["metrics"].items()}}))
    env.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
