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


# --------------------------------------------------------------------------------------------- robot-env layer, per robot
# `_get_obs` / `_get_info` / `initial_qpos` of the four robot envs restated over an oracle's (qpos, qvel, sensors).  The joint
# name lists are the reference's own; addresses follow MuJoCo's document order (one free joint, then one hinge per joint
# element: qpos address 7 + k, dof address 6 + k for the k-th hinge), read from the MJCF text -- not from cosim_amd.compile.
_W4_LEGS = ("FL", "FR", "RL", "RR")
ROBOT_ENV = {
    # flamingo_light_v1.py:95-98,171,229 (no gear; info state = [dof_pos[0], dof_pos[1], dof_vel[2], dof_vel[3]])
    "flamingo_light_v1": dict(
        qpos_names=["left_shoulder_joint", "right_shoulder_joint"],
        qvel_names=["left_shoulder_joint", "right_shoulder_joint", "left_wheel_joint", "right_wheel_joint"],
        geared=slice(0, 0), gear=1.0, info_state=[("pos", 0), ("pos", 1), ("vel", 2), ("vel", 3)],
        init_height=0.13, init_noise=["left_shoulder_joint", "right_shoulder_joint", "left_wheel_joint", "right_wheel_joint"],
        action_scale=lambda hw: [hw["action_scales"]["shoulder"]] * 2 + [hw["action_scales"]["wheel"]] * 2),
    # flamingo_p_v3.py:107-120 (dof_pos[4:6], dof_vel[4:6] times gear_ratio), :207 (info state ungeared), :248-254
    "flamingo_p_v3": dict(
        qpos_names=["left_hip_joint", "right_hip_joint", "left_shoulder_joint", "right_shoulder_joint", "left_leg_joint", "right_leg_joint"],
        qvel_names=["left_hip_joint", "right_hip_joint", "left_shoulder_joint", "right_shoulder_joint", "left_leg_joint", "right_leg_joint",
                    "left_wheel_joint", "right_wheel_joint"],
        geared=slice(4, 6), gear="gear_ratio", info_state=[("pos", i) for i in range(6)] + [("vel", 6), ("vel", 7)],
        init_height=0.61282, init_noise="all",
        action_scale=lambda hw: [hw["action_scales"][k] for k in ("hip", "hip", "shoulder", "shoulder", "leg", "leg", "wheel", "wheel")]),
    # w4_p_v2.py:107-120 (dof_pos[8:12], dof_vel[8:12] times gear_ratio), :204-206, :243-249
    "w4_p_v2": dict(
        qpos_names=[f"{l}_{n}_joint" for n in ("hip", "shoulder", "leg") for l in _W4_LEGS],
        qvel_names=[f"{l}_{n}_joint" for n in ("hip", "shoulder", "leg", "wheel") for l in _W4_LEGS],
        geared=slice(8, 12), gear="gear_ratio", info_state=[("pos", i) for i in range(12)] + [("vel", i) for i in range(12, 16)],
        init_height=0.47957, init_noise="all",
        action_scale=lambda hw: [hw["action_scales"][n] for n in ("hip", "shoulder", "leg", "wheel") for _ in _W4_LEGS]),
    # humanoid_p_v0.py:139-153 (joint_names_in_order), :268 (info state = dof_pos), :305-311
    "humanoid_p_v0": dict(
        qpos_names=["left_hip_pitch_joint", "right_hip_pitch_joint", "torso_joint", "left_hip_roll_joint", "right_hip_roll_joint",
                    "left_shoulder_pitch_joint", "right_shoulder_pitch_joint", "left_hip_yaw_joint", "right_hip_yaw_joint",
                    "left_shoulder_roll_joint", "right_shoulder_roll_joint", "left_knee_joint", "right_knee_joint",
                    "left_shoulder_yaw_joint", "right_shoulder_yaw_joint", "left_ankle_pitch_joint", "right_ankle_pitch_joint",
                    "left_elbow_pitch_joint", "right_elbow_pitch_joint", "left_ankle_roll_joint", "right_ankle_roll_joint",
                    "left_elbow_yaw_joint", "right_elbow_yaw_joint"],
        qvel_names=None,   # same list (:153)
        geared=slice(0, 0), gear=1.0, info_state=[("pos", i) for i in range(23)],
        init_height=1.105, init_noise="all", action_scale=None),
}


def hinge_addresses(xml_path: str) -> Dict[str, int]:
    """joint name -> index k of the hinge in document order (qpos address 7 + k, dof address 6 + k)."""
    import re
    names = re.findall(r'<joint\s+name="([A-Za-z0-9_]+)"', open(xml_path).read())
    assert names and "free" in names[0], "first joint of the cosim robots is the free joint"
    return {n: k for k, n in enumerate(names[1:])}


class RobotEnvOracle:
    """What ``<Robot>._get_obs`` / ``_get_info`` read from ``self.data`` (noise excluded), given an ``Oracle``."""

    def __init__(self, env_id: str, xml_path: str, hardware: dict):
        self.r = ROBOT_ENV[env_id]
        adr = hinge_addresses(xml_path)
        qn = self.r["qpos_names"]
        vn = self.r["qvel_names"] or qn
        self.q_idx = np.array([7 + adr[n] for n in qn])
        self.qd_idx = np.array([6 + adr[n] for n in vn])
        g = self.r["gear"]
        self.gear = float(hardware[g]) if isinstance(g, str) else float(g)
        self.nhinge = len(adr)
        self.noise_qadr = (np.arange(7, 7 + self.nhinge) if self.r["init_noise"] == "all"
                           else np.array([7 + adr[n] for n in self.r["init_noise"]]))
        sc = self.r["action_scale"]
        self.action_scale = None if sc is None else np.array(sc(hardware), dtype=np.float64)

    def obs(self, o, action, height_map=None) -> dict:
        dof_pos, dof_vel = o.qpos[self.q_idx].copy(), o.qvel[self.qd_idx].copy()
        s = self.r["geared"]
        dof_pos[s] *= self.gear
        dof_vel[s] *= self.gear
        out = {"dof_pos": dof_pos, "dof_vel": dof_vel, "ang_vel": o.sensor_gyro.copy(), "lin_vel": o.sensor_vel.copy(),
               "projected_gravity": projected_gravity(o.sensor_quat), "last_action": np.asarray(action, dtype=np.float64)}
        if height_map is not None:
            out["height_map"] = height_map
        return out

    def info(self, o, action, prev_action, torque) -> dict:
        dof_pos, dof_vel = o.qpos[self.q_idx], o.qvel[self.qd_idx]
        a, p = np.asarray(action, dtype=np.float64), np.asarray(prev_action, dtype=np.float64)
        return {"action_diff_RMSE": float(np.sqrt(np.mean((a - p) ** 2))), "torque": np.asarray(torque, dtype=np.float64),
                "lin_vel_x": float(o.sensor_vel[0]), "lin_vel_y": float(o.sensor_vel[1]), "ang_vel_yaw": float(o.sensor_gyro[2]),
                "state": np.array([dof_pos[i] if k == "pos" else dof_vel[i] for k, i in self.r["info_state"]])}
