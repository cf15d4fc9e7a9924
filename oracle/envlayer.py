"""numpy restatement of the reference's pure-Python per-step pieces.  TEST INFRASTRUCTURE.

Each function/class cites the reference lines it follows and is pinned by the golden vectors that
``tools/make_golden.py`` captured from the reference's own modules (``tests/golden/``, checked in
``tests/test_golden_oracle.py``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np


def pd_controller(kp, tq, q, kd, td, d):
    """``ControlManager.pd_controller`` (reference manager/control_manager.py:11-12)."""
    return kp * (tq - q) + kd * (td - d)


def delay_filter_sequence(actions: np.ndarray, u: np.ndarray, prob: float) -> np.ndarray:
    """``ControlManager.delay_filter`` applied to a sequence (control_manager.py:14-23).

    ``u[t]`` is the uniform draw of call ``t``; delayed when ``prob > u[t]`` and a previous action exists.
    """
    out = np.empty_like(actions)
    prev = None
    for t, a in enumerate(actions):
        delay = prob > u[t]
        if not delay or prev is None:
            out[t] = a
        else:
            out[t] = prev
        prev = a
    return out


def projected_gravity(quat_wxyz: np.ndarray) -> np.ndarray:
    """``MathUtils.quat_to_base_vel(quat_xyzw, [0,0,-1])`` (flamingo_light_v1.py:105-108; utils/math_utils.py:41-44):
    scipy normalises the quaternion and applies the inverse rotation, i.e. ``R(q)^T [0,0,-1]``."""
    q = np.asarray(quat_wxyz, dtype=np.float64)
    if np.all(q == 0):
        q = np.array([1.0, 0, 0, 0])
    w, x, y, z = q / np.linalg.norm(q)
    # third row of R, negated
    return -np.array([2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z])


def rot_matrix_wxyz(q) -> np.ndarray:
    """``MathUtils.quat_to_rot_matrix`` (utils/math_utils.py:47-52) — does NOT normalise."""
    w, x, y, z = q
    return np.array([[1 - 2 * y ** 2 - 2 * z ** 2, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                     [2 * x * y + 2 * z * w, 1 - 2 * x ** 2 - 2 * z ** 2, 2 * y * z - 2 * x * w],
                     [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x ** 2 - 2 * y ** 2]])


def push_velocity(qpos: np.ndarray, v_world) -> np.ndarray:
    """``event('push', v)`` (flamingo_light_v1.py:234-243): returns the new ``qvel[0:3]``."""
    R = rot_matrix_wxyz(np.asarray(qpos[3:7], dtype=np.float64)).T
    v = np.asarray(v_world, dtype=np.float64).reshape(3)
    r = R @ v
    return np.array([r[0], r[1], v[2]])


class WrapperOracle:
    """``CommandWrapper(TimeLimitWrapper(StateBuildWrapper(env)))`` restated over plain observation dicts
    (reference envs/wrappers.py:88-417).  ``obs`` dicts are what ``<Robot>._get_obs`` returns."""

    def __init__(self, config: dict, obs_to_dim: Dict[str, int], control_freq: float = 50.0):
        ob = config["observation"]
        self.config = config
        self.control_freq = float(control_freq)
        self.command_dim = ob["command_dim"]
        self.stack_size = int(ob["stack_size"])
        self.stacked = list(ob["stacked_obs_order"])
        self.non_stacked = list(ob["non_stacked_obs_order"])
        self.dims = obs_to_dim
        self.stacked_dim = sum(obs_to_dim[n] for n in self.stacked)
        self.non_stacked_dim = sum(obs_to_dim[n] for n in self.non_stacked)
        self.state_dim = self.stack_size * self.stacked_dim + self.non_stacked_dim
        self.cmd_slices = self._cmd_slices()
        self.obs_buffer = np.zeros((self.stack_size, self.stacked_dim), dtype=np.float32)
        self.cache: Dict[str, np.ndarray] = {}
        self.sim_step = 0
        self.max_sim_step = int(config["env"]["max_duration"] * self.control_freq)
        self.applied_command = np.zeros(self.command_dim)
        self.user_command = np.zeros(self.command_dim)

    def _cmd_slices(self) -> List[slice]:  # wrappers.py:129-158
        out: List[slice] = []
        if self.command_dim <= 0:
            return out
        off, starts = 0, []
        for n in self.stacked:
            if n == "command":
                starts.append(off)
            off += self.dims[n]
        for k in range(self.stack_size):
            for s in starts:
                out.append(slice(k * self.stacked_dim + s, k * self.stacked_dim + s + self.command_dim))
        base, off = self.stack_size * self.stacked_dim, 0
        for n in self.non_stacked:
            if n == "command":
                out.append(slice(base + off, base + off + self.command_dim))
            off += self.dims[n]
        return out

    def _concat(self, obs: dict, names: List[str]) -> np.ndarray:  # wrappers.py:160-202
        parts = []
        for n in names:
            if n == "command":
                parts.append(np.zeros((self.command_dim,), dtype=np.float32))
                continue
            cfg = self.config["observation"][n]
            freq, scale = float(cfg["freq"]), float(cfg["scale"])
            if freq <= 0:
                raise ValueError(f"Invalid observation update frequency for '{n}': {freq}. Must be > 0.")
            interval = max(1, int(round(self.control_freq / freq)))
            if self.sim_step == 0 or self.sim_step % interval == 0 or n not in self.cache:
                self.cache[n] = np.asarray(obs[n], dtype=np.float32) * scale
            parts.append(self.cache[n].ravel().astype(np.float32))
        return np.concatenate(parts, axis=0) if parts else np.zeros((0,), dtype=np.float32)

    def _build(self, obs: dict, reset: bool) -> np.ndarray:  # wrappers.py:204-243
        frame = self._concat(obs, self.stacked)
        if reset:
            self.obs_buffer[:] = frame
        else:
            if self.stack_size > 1:
                self.obs_buffer[1:, :] = self.obs_buffer[:-1, :].copy()
            self.obs_buffer[0, :] = frame
        state = np.concatenate([self.obs_buffer.ravel(), self._concat(obs, self.non_stacked)], axis=0).astype(np.float32)
        for s in self.cmd_slices:  # _apply_command_inplace, wrappers.py:377-383
            state[s] = self.applied_command
        return state

    def receive_user_command(self, user_command, qpos: Optional[np.ndarray] = None):  # wrappers.py:349-375
        user_command = np.asarray(user_command, dtype=np.float64)
        self.user_command = user_command[:self.command_dim]
        self.applied_command = np.array(user_command[:self.command_dim], dtype=np.float64)
        if self.config["env"]["position_command"] is False:
            for i in range(self.command_dim):
                self.applied_command[i] *= self.config["observation"]["command_scales"][str(i)]
        else:
            assert self.command_dim == 2
            dx, dy = self.user_command[0] - qpos[0], self.user_command[1] - qpos[1]
            w, x, y, z = np.asarray(qpos[3:7], dtype=np.float64)
            yaw = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
            c, s = np.cos(-yaw), np.sin(-yaw)
            self.applied_command[0] = c * dx - s * dy
            self.applied_command[1] = s * dx + c * dy

    def reset(self, obs: dict) -> np.ndarray:  # wrappers.py:245-256,303-307,385-389
        self.sim_step = 0
        self.cache.clear()
        return self._build(obs, reset=True)

    def step(self, obs: dict, terminated: bool = False):  # wrappers.py:258-269,309-320,391-405
        self.sim_step += 1
        state = self._build(obs, reset=False)
        truncated = self.sim_step == self.max_sim_step
        return state, terminated, truncated
