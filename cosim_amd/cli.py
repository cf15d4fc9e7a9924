"""Headless runner: the reference's GUI + ``Tester`` session as one command.

    python -m cosim_amd.cli --env flamingo_light_v1 --num-envs 4096 --steps 1000 --command 0.5 0 0 0 \\
        --policy sinusoid | random-mlp | path/to/actor.onnx  [--terrain rocky_hard] [--push-at 200 --push 0.5 0 0] \\
        [--report report.json] [--trace-env 0]
    python -m cosim_amd.cli --config session.yaml

One process per GPU: under ``torchrun`` every rank simulates its shard of ``--num-envs`` and rank 0 writes the report.

``--config`` (YAML in, report out) replaces a GUI session of the reference: what ``ui/main_window.py:709-788`` gathers from
widgets, plus the two things a user does WHILE the test runs -- key-driven commands (``ui/main_window.py:272-290`` ->
``Tester.update_command``, core/tester.py:41-46) as a time series, and the push button (held: ``core/tester.py:80-81`` applies
the push every loop iteration while ``_push_event`` is set) as a schedule:

    env:      {id: flamingo_p_v3, terrain: rocky_easy, max_duration: 120.0, position_command: false}
    engine:   {num_envs: 4096, seed: 1234}
    random:   {sensor_noise: low, action_delay_prob: 0.05, ...}        # overrides of the GUI defaults (config.make_config)
    observation: {stack_size: 3, ...}                                   # likewise
    policy:   {kind: sinusoid | random-mlp | onnx, onnx_file: actor.onnx, use_lstm: false, h_in_dim: 256, c_in_dim: 256}
    steps:    1000
    commands: [[0, 0.5, 0, 0, 0], [200, 1.0, 0, 0.3, 0]]              # from control step t on: user_command = c0 .. c3
    pushes:   [[300, 310, 0.5, 0, 0]]                                   # held for steps t0 <= k < t1: event("push", [vx, vy, vz])
    report:   report.json
    trace_env: 0
    percentiles: true

Flags given on the command line override the file.
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
    ap.add_argument("--config", default="", help="YAML session file (see the module docstring); flags override it")
    ap.add_argument("--env", default=None)
    ap.add_argument("--terrain", default=None)
    ap.add_argument("--num-envs", type=int, default=None, help="total over all ranks")
    ap.add_argument("--steps", type=int, default=None, help="control steps (50 Hz)")
    ap.add_argument("--policy", default=None, help="sinusoid | random-mlp | <file.onnx>")
    ap.add_argument("--lstm", action="store_true", help="the ONNX file is an LSTM policy with h_in / c_in inputs")
    ap.add_argument("--hidden-dim", type=int, default=256, help="h_in_dim = c_in_dim of an LSTM policy")
    ap.add_argument("--command", type=float, nargs="*", default=None)
    ap.add_argument("--position-command", action="store_true")
    ap.add_argument("--max-duration", type=float, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--push-at", type=int, default=-1, help="control step at which a push event fires")
    ap.add_argument("--push", type=float, nargs=3, default=[0.5, 0.0, 0.0])
    ap.add_argument("--report", default=None, help="write the fleet report (JSON) here")
    ap.add_argument("--trace-env", type=int, default=None, help="also keep the per-step info series of this local env")
    ap.add_argument("--percentiles", action="store_true", help="p5 / p50 / p95 of tracking error, |torque| and action-RMSE in the report")
    ap.add_argument("--graph", action="store_true", help="capture policy -> step -> report in a HIP graph and replay it (ONNX policies)")
    ap.add_argument("--pipelined", action="store_true",
                    help="policy -> step -> report per env range on the range's own stream, no fleet-wide barrier per step (ONNX MLP policies)")
    ap.add_argument("--ranges", type=int, default=4, help="env ranges of --pipelined")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args(argv)

    # ---- session file: the GUI's config dict + the user's key / push input over time
    sess = {}
    if args.config:
        import yaml
        with open(args.config) as f:
            sess = yaml.safe_load(f) or {}
        if not isinstance(sess, dict):
            ap.error("--config: the YAML document must be a mapping")
        unknown = set(sess) - {"env", "engine", "random", "observation", "hardware", "policy", "steps", "commands", "pushes", "report",
                               "trace_env", "percentiles"}
        if unknown:
            ap.error(f"--config: unknown top-level keys {sorted(unknown)}")
    s_env, s_eng, s_pol = sess.get("env", {}) or {}, sess.get("engine", {}) or {}, sess.get("policy", {}) or {}

    def pick(flag, file_value, default):
        return flag if flag is not None else (file_value if file_value is not None else default)
    args.env = pick(args.env, s_env.get("id"), "flamingo_light_v1")
    args.terrain = pick(args.terrain, s_env.get("terrain"), "flat")
    args.max_duration = float(pick(args.max_duration, s_env.get("max_duration"), 120.0))
    args.position_command = bool(args.position_command or s_env.get("position_command", False))
    args.num_envs = int(pick(args.num_envs, s_eng.get("num_envs"), 1024))
    args.seed = int(pick(args.seed, s_eng.get("seed"), 1234))
    args.steps = int(pick(args.steps, sess.get("steps"), 500))
    kind = s_pol.get("kind")
    args.policy = pick(args.policy, s_pol.get("onnx_file") if kind == "onnx" else kind, "sinusoid")
    args.lstm = bool(args.lstm or s_pol.get("use_lstm", False))
    if s_pol.get("h_in_dim") is not None:
        args.hidden_dim = int(s_pol["h_in_dim"])
    args.report = pick(args.report, sess.get("report"), "")
    args.trace_env = int(pick(args.trace_env, sess.get("trace_env"), -1))
    args.percentiles = bool(args.percentiles or sess.get("percentiles", False))
    # command time series: rows [t, c0, c1, ...]; --command is the row [0, c...]
    commands = [[float(x) for x in row] for row in (sess.get("commands") or [])]
    if args.command is not None:
        commands = [[0.0] + list(args.command)]
    if not commands:
        commands = [[0.0, 0.5, 0.0, 0.0, 0.0]]
    commands.sort(key=lambda r: r[0])
    pushes = [[float(x) for x in row] for row in (sess.get("pushes") or [])]
    if args.push_at >= 0:
        pushes.append([float(args.push_at), float(args.push_at + 1)] + [float(x) for x in args.push])
    for row in pushes:
        if len(row) != 5 or row[1] <= row[0]:
            ap.error("pushes: rows are [t0, t1, vx, vy, vz] with t1 > t0")

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
    for section in ("random", "observation", "hardware"):          # the widgets' values (ui/main_window.py:750-787)
        for k, v in (sess.get(section) or {}).items():
            if k not in cfg[section]:
                ap.error(f"--config: unknown key {section}.{k}")
            if isinstance(cfg[section][k], dict) and isinstance(v, dict):
                cfg[section][k].update(v)
            else:
                cfg[section][k] = v
    env = BatchedEnv(cfg, num_envs=hi - lo, device=dev, seed=args.seed, auto_reset=True, env_id0=lo,
                     **({"ranges": args.ranges, "deferred_join": True} if args.pipelined else {}))
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
    if args.graph and (pushes or len(commands) > 1 or args.trace_env >= 0):
        ap.error("--graph replays one captured control step: pushes, command changes and --trace-env need the eager loop")
    if args.pipelined and (not hasattr(policy, "get_action_into") or args.graph or args.trace_env >= 0):
        ap.error("--pipelined needs an ONNX MLP policy (random-mlp or a file) and excludes --graph / --trace-env")
    rep = FleetReporter(env, trace_env=args.trace_env if args.trace_env >= 0 else None, percentiles=args.percentiles)
    run = Runner(env, policy, reporter=rep)
    for i, v in enumerate(commands[0][1:1 + env.command_dim]):
        run.update_command(i, v)

    def before_step(k):
        """What the reference's UI thread does between two loop iterations: key-driven command changes and the push button."""
        for row in commands:
            if int(row[0]) == k:
                for i, v in enumerate(row[1:1 + env.command_dim]):
                    run.update_command(i, v)                       # tester.py:41-46
        held = [row for row in pushes if row[0] <= k < row[1]]
        if held:
            run.activate_push_event(np.asarray(held[-1][2:5], dtype=np.float32))   # tester.py:48-50; applied while held (:80-81)
        else:
            run.deactivate_push_event()

    # episodes ended are counted on the device (engine meta word 11) and read once after the run: a per-step `.item()` on the done
    # flags would drain the GPU queue at every control step
    on_step = None
    episodes0 = None
    torch.cuda.synchronize(env.device)
    episodes0 = env.solver_stats()["episodes_ended"]
    t0 = time.perf_counter()
    if args.pipelined:
        if pushes or len(commands) > 1:
            ap.error("--pipelined runs one command and no push schedule (use Runner.test_pipelined with update_command for more)")
        n = run.test_pipelined(args.steps)
    else:
        n = run.test_graphed(args.steps) if args.graph else run.test(max_steps=args.steps, on_step=on_step, before_step=before_step)
    torch.cuda.synchronize(env.device)
    dt = time.perf_counter() - t0
    rep.episodes_ended = env.solver_stats()["episodes_ended"] - episodes0
    out = rep.save(args.report) if (args.report and rank == 0) else rep.summary()
    if rank == 0:
        print(json.dumps({"env": args.env, "terrain": args.terrain, "envs_total": args.num_envs, "ranks": world, "control_steps": n,
                          "env_steps_per_s_this_rank": env.num_envs * n / dt, "episodes_ended": out["episodes_ended"],
                          "metrics": {k: round(v["mean"], 5) for k, v in out["metrics"].items()},
                          **({"percentiles": {k: {q: round(x, 5) for q, x in v.items()} for k, v in out["percentiles"].items()}}
                             if "percentiles" in out else {})}))
    env.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
