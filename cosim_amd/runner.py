"""Headless batched counterpart of ``Tester.test`` (reference ``core/tester.py:55-103``).

Same sequencing as the reference loop — ``reset``; then per control step ``receive_user_command`` -> ``policy.get_action``
-> optional ``event("push")`` -> ``step`` -> ``reporter.write_info`` — without Qt, GLFW or ONNX Runtime, over N envs.
``policy`` is anything with ``get_action(state) -> action`` (the contract of ``core/policy.py:11-21,34-47``: float
actions clipped to [-1, 1]); ``reporter`` anything with ``write_info(info)`` (``core/reporter.py:210-218``).  For the
reference's single-env ``Reporter`` pass ``report_env=i``: the batched ``info`` is sliced to env ``i`` with numpy / Python
scalars, i.e. exactly the dict ``Reporter`` expects.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


class SinusoidPolicy:
    """Synthetic stand-in for the ONNX policy (none ships with the reference): the BASELINE.md drive
    ``a[n,j,t] = clip(0.25 sin(2 pi 0.5 Hz 0.02 t + phi[n,j]))`` with Philox phases keyed by global env id."""

    def __init__(self, num_envs: int, action_dim: int, device, env_id0: int = 0, amplitude: float = 0.25, seed: int = 1234):
        import torch
        from . import rng as crng
        gids = np.arange(env_id0, env_id0 + num_envs, dtype=np.uint64)[:, None]
        phi = 2 * np.pi * crng.uniform(seed, gids, 0, 5, np.arange(action_dim)[None, :]).astype(np.float32)
        self.phi = torch.tensor(phi, device=device)
        self.amplitude, self.t, self.torch = amplitude, 0, torch

    def get_action(self, state):
        a = self.amplitude * self.torch.sin(2 * np.pi * 0.5 * 0.02 * self.t + self.phi)
        self.t += 1
        return a.clamp_(-1.0, 1.0)


class Runner:
    """``Tester`` without the GUI: ``load_config`` / ``update_command`` / ``activate_push_event`` / ``test`` / ``stop``."""

    def __init__(self, env, policy, reporter=None, report_env: Optional[int] = None):
        self.env, self.policy, self.reporter, self.report_env = env, policy, reporter, report_env
        self.user_command = np.zeros(max(env.command_dim, 0), dtype=np.float32)
        self._push_event, self._push_vel, self._stop = False, None, False

    def update_command(self, index: int, value: float):            # tester.py:41-46
        if index < self.env.command_dim:
            self.user_command[index] = value

    def activate_push_event(self, push_vel):                       # tester.py:48-50
        self._push_event, self._push_vel = True, push_vel

    def deactivate_push_event(self):
        self._push_event = False

    def stop(self):
        self._stop = True

    def _one_env_info(self, info: dict, i: int) -> dict:
        out = {}
        for k, v in info.items():
            if hasattr(v, "ndim") and getattr(v, "ndim", 0) >= 1 and v.shape[0] == self.env.num_envs:
                x = v[i].detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v[i])
                out[k] = x.astype(np.float64) if x.ndim else float(x)
            else:
                out[k] = v
        return out

    def test(self, max_steps: Optional[int] = None, on_step: Optional[Callable] = None, before_step: Optional[Callable] = None) -> int:
        """Run until every env is done (no auto-reset) or ``max_steps`` control steps (auto-reset); returns steps run.
        ``before_step(k)`` runs at the top of iteration k, where the reference's UI thread has written the key-driven command and
        push flags the loop is about to read (core/tester.py:39,80: ``user_command`` / ``_push_event``)."""
        env = self.env
        state, _ = env.reset()
        steps = 0
        done_all, done_seen = False, None
        while not done_all and not self._stop and (max_steps is None or steps < max_steps):
            if before_step is not None:
                before_step(steps)
            env.receive_user_command(self.user_command)            # tester.py:68
            action = self.policy.get_action(state)                 # :70
            if self._push_event:
                env.event("push", self._push_vel)                  # :80-81
            state, terminated, truncated, info = env.step(action)  # :90
            if env.auto_reset and hasattr(self.policy, "reset"):
                self.policy.reset(terminated | truncated)          # the reference builds a fresh policy per episode (tester.py:57-60)
            if self.reporter is not None:
                self.reporter.write_info(self._one_env_info(info, self.report_env) if self.report_env is not None else info)
            if on_step is not None:
                on_step(steps, state, terminated, truncated, info)
            steps += 1
            if not env.auto_reset:
                # every env has ended an episode at least once (they need not end in the same step: a non-finite state
                # restarts one env early)
                done_seen = (terminated | truncated) if done_seen is None else (done_seen | terminated | truncated)
                done_all = bool(done_seen.all().item())
        return steps

    def test_graphed(self, max_steps: int, warmup: int = 3) -> int:
        """The same loop with one control step (policy -> step -> reporter update) captured in a HIP graph and replayed:
        the per-step host work of ~40 small launches collapses into one graph launch.  Needs an auto-reset env, a policy
        whose ``get_action`` is pure device work on persistent tensors (``policy.MLPPolicy``; not ``SinusoidPolicy``, whose
        clock lives on the host) and a reporter that only updates device buffers (``reporter.FleetReporter`` without
        ``trace_env``).  The user command and push flag are read from persistent device tensors, so ``update_command``
        between replays still takes effect; a push is applied outside the graph."""
        env, t = self.env, self.env.torch
        if not env.auto_reset:
            raise ValueError("test_graphed needs auto_reset=True (no per-step host check of the done flags)")
        if not getattr(self.policy, "graph_safe", False):
            raise ValueError(f"test_graphed: {type(self.policy).__name__}.get_action is not pure device work on persistent tensors "
                             "(a host-side clock or state would be frozen into the captured graph); use an ONNX policy "
                             "(policy.MLPPolicy / policy.LSTMPolicy) or the eager Runner.test loop")
        if self.reporter is not None and getattr(self.reporter, "trace_env", None) is not None:
            raise ValueError("test_graphed: a reporter with trace_env reads the device every step; use Runner.test")
        state, _ = env.reset()
        # the device-side episode counter is cumulative since the engine was created (earlier runs on a reused env included): this
        # run reports the difference, like the eager loop, which counts per run
        episodes0 = env.solver_stats()["episodes_ended"]
        env.receive_user_command(self.user_command)
        action = t.zeros((env.num_envs, env.action_dim), dtype=t.float32, device=env.device)

        def one_step():
            action.copy_(self.policy.get_action(env.state))
            _, terminated, truncated, info = env.step(action)
            if hasattr(self.policy, "reset"):
                self.policy.reset(terminated | truncated)          # device-side mask: replayed with the graph
            if self.reporter is not None:
                self.reporter.write_info(info)
        side = t.cuda.Stream(device=env.device)
        side.wait_stream(t.cuda.current_stream(env.device))
        with t.cuda.stream(side):
            for _ in range(warmup):
                one_step()
        t.cuda.current_stream(env.device).wait_stream(side)
        t.cuda.synchronize(env.device)
        graph = t.cuda.CUDAGraph()
        with t.cuda.graph(graph):
            one_step()
        steps = warmup                            # capturing records the step, it does not run it
        while steps < max_steps and not self._stop:
            env.receive_user_command(self.user_command)
            if self._push_event:
                env.event("push", self._push_vel)
            graph.replay()
            steps += 1
        if self.reporter is not None and hasattr(self.reporter, "steps"):
            self.reporter.steps = steps          # write_info ran once per replay on the device, once in Python
        if self.reporter is not None and hasattr(self.reporter, "episodes_ended"):
            self.reporter.episodes_ended = env.solver_stats()["episodes_ended"] - episodes0   # counted on the device (meta[11]), graph or not
        return steps
