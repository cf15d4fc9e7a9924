"""``build_env(config)`` — the drop-in factory (reference ``envs/build.py:8-24``).

``config["engine"]["num_envs"] == 1`` (or no ``engine`` key) returns :class:`SingleEnv`, which
presents *exactly* the reference's single-environment API — numpy in, numpy out, Python bools,
the same ``info`` keys, the same assertions — so ``core/tester.py``-style loops, ``core/policy.py``
and ``core/reporter.py`` work unchanged.  ``num_envs > 1`` returns the batched object.
"""
from __future__ import annotations

import numpy as np

from .batched_env import BatchedEnv
from .robots import ROBOTS


class SingleEnv:
    """N = 1 adapter over :class:`BatchedEnv` with the reference's ``BaseEnv`` semantics (wrappers.py:8-85)."""

    def __init__(self, config: dict):
        self.env = BatchedEnv(config, num_envs=1, auto_reset=False)
        e = self.env
        self.config = config
        self.id, self.action_dim, self.state_dim = e.id, e.action_dim, e.state_dim
        self.command_dim, self.cmd_slices = e.command_dim, e.cmd_slices
        self.control_freq, self.obs_to_dim = e.control_freq, e.obs_to_dim
        self.max_sim_step = e.max_sim_step
        self.user_command = np.zeros(config["observation"]["command_dim"])
        self.reset_flag = False

    def receive_user_command(self, user_command):
        user_command = np.asarray(user_command, dtype=np.float64)
        self.user_command = user_command[:self.command_dim]
        self.env.receive_user_command(np.asarray(user_command[:max(self.command_dim, 0)], dtype=np.float32))

    def _info(self, info: dict, first: bool) -> dict:
        out = {"dt": info["dt"]}
        nu = self.action_dim
        buf = self.env.info_buf[0].cpu().numpy().astype(np.float64)
        out["action"] = np.zeros(nu) if first else np.asarray(info["action"][0].cpu().numpy(), dtype=np.float64).copy()
        out["action_diff_RMSE"] = float(buf[0]) if not first else 0.0
        out["torque"] = buf[4:4 + nu].copy() if not first else np.zeros(nu)
        out["lin_vel_x"], out["lin_vel_y"], out["ang_vel_yaw"] = (np.float32(buf[1]), np.float32(buf[2]), float(buf[3])) \
            if not first else (np.float32(0), np.float32(0), 0.0)
        out["set_points"] = buf[4 + nu:4 + 2 * nu].copy() if not first else np.zeros(nu)
        if first:
            d = self.env.get_data()
            from .model import get_field
            blob = self.env.cm.blob
            qp, qv = d.qpos[0].cpu().numpy(), d.qvel[0].cpu().numpy()
            kinds, adrs, gears = (np.array(get_field(blob, k)[:blob.ninfo_state]) for k in ("info_kind", "info_adr", "info_gear"))
            out["state"] = [float((qp[a] if k == 0 else qv[a]) * g) for k, a, g in zip(kinds, adrs, gears)]
        else:
            out["state"] = [float(x) for x in buf[4 + 2 * nu:]]
        return out

    def reset(self):
        self.reset_flag = True
        state, _ = self.env.reset()
        info = self._info({"dt": self.env.dt_ * self.env.frame_skip}, first=True)
        return state[0].cpu().numpy().astype(np.float32), info

    def step(self, action: np.ndarray):
        assert self.reset_flag is True, "Call 'reset()' before calling 'step()'."
        action = np.asarray(action)
        if action.shape != (self.action_dim,):
            # gymnasium MujocoEnv.do_simulation raises ValueError on a ctrl shape mismatch
            raise ValueError(f"Action dimension mismatch. Expected {(self.action_dim,)}, found {action.shape}")
        state, term, trunc, info = self.env.step(np.asarray(action, dtype=np.float32)[None, :])
        terminated, truncated = bool(term[0].item()), bool(trunc[0].item())
        out = self._info(info, first=False)
        for i in range(self.command_dim):
            out[f"user_command_{i}"] = self.user_command[i]
        if terminated or truncated:
            self.reset_flag = False
        return state[0].cpu().numpy().astype(np.float32), terminated, truncated, out

    def event(self, event: str, value):
        return self.env.event(event, np.asarray(value, dtype=np.float32).reshape(3,))

    def get_data(self):
        d = self.env.get_data()

        class _D:
            qpos = d.qpos[0].cpu().numpy().astype(np.float64)
            qvel = d.qvel[0].cpu().numpy().astype(np.float64)
        return _D()

    def render(self):
        pass

    def close(self):
        self.env.close()


def build_env(config: dict):
    if config["env"]["id"] not in ROBOTS:
        raise NameError(f"Please select a valid environment id. Received '{config['env']['id']}'.")
    n = int(config.get("engine", {}).get("num_envs", 1))
    if n == 1:
        return SingleEnv(config)
    return BatchedEnv(config)
