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

    def test_pipelined(self, max_steps: int, report_every: int = 1, inflight: int = 2) -> int:
        """The same loop -- policy -> step -> reporter, ``core/tester.py:66-97`` -- with NO fleet-wide barrier per step: the env was
        built with ``ranges=S`` (S > 1), and each range of envs runs its own chain policy(range) -> step(range) -> reporter(range) on
        the range's own stream (``BatchedEnv.range_streams``).  A range's next control step then fills the tail of the others'
        launches, as in the bench; the policy only ever needs the states of its own range.  Needs an auto-reset env, a stateless
        device policy with ``get_action_into`` (``policy.MLPPolicy``) and a ``FleetReporter`` (or none).  The user command and a held
        push are read before each step like in ``test`` (changing them joins the ranges first).  ``inflight``: how many steps the host
        may run ahead of each range (deep queues step slower)."""
        env, t = self.env, self.env.torch
        if len(env.range_list) < 2:
            raise ValueError("test_pipelined needs an env built with ranges > 1")
        if not env.auto_reset:
            raise ValueError("test_pipelined needs auto_reset=True (no per-step host check of the done flags)")
        if not hasattr(self.policy, "get_action_into") or not getattr(self.policy, "graph_safe", False):
            raise ValueError(f"test_pipelined: {type(self.policy).__name__} cannot be evaluated per range on a stream of its own "
                             "(needs get_action_into and no per-env or host-side state); use Runner.test")
        if self.reporter is not None and not hasattr(self.reporter, "write_info_range"):
            raise ValueError("test_pipelined: the reporter must reduce per range (reporter.FleetReporter)")
        env.reset()
        episodes0 = env.solver_stats()["episodes_ended"]
        env.receive_user_command(self.user_command)
        last_cmd = self.user_command.copy()
        action = t.zeros((env.num_envs, env.action_dim), dtype=t.float32, device=env.device)
        cur = t.cuda.current_stream(env.device)
        for st in env.range_streams:
            st.wait_stream(cur)                                    # the reset, the command upload and `action` are ready
        ring = [[t.cuda.Event() for _ in range(max(inflight, 1))] for _ in env.range_list]
        steps = 0
        while steps < max_steps and not self._stop:
            if self._push_event or not np.array_equal(last_cmd, self.user_command):
                env.join()                                         # whole-fleet inputs change: the ranges meet here
                t.cuda.current_stream(env.device).synchronize()
                env.receive_user_command(self.user_command)
                last_cmd = self.user_command.copy()
                if self._push_event:
                    env.event("push", self._push_vel)
                cur = t.cuda.current_stream(env.device)
                for st in env.range_streams:
                    st.wait_stream(cur)
            sample = self.reporter is not None and steps % report_every == 0
            for i, (first, count) in enumerate(env.range_list):
                if inflight > 0 and steps >= inflight:
                    ring[i][steps % inflight].synchronize()
                with t.cuda.stream(env.range_streams[i]):
                    self.policy.get_action_into(env.state[first:first + count], action[first:first + count])   # tester.py:70
                    env.step_range(first, count, action)                                                          # :90
                    if sample:
                        self.reporter.write_info_range(first, count)                                              # :92
                    if inflight > 0:
                        ring[i][steps % inflight].record()
                env.range_mark(i)
            steps += 1
        env.join()
        t.cuda.current_stream(env.device).synchronize()
        if self.reporter is not None and hasattr(self.reporter, "episodes_ended"):
            self.reporter.episodes_ended = env.solver_stats()["episodes_ended"] - episodes0
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
